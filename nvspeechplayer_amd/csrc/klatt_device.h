// klatt_device.h -- the Klatt synthesis kernel for gfx950 (MI355X, CDNA4).
//
// One utterance (one speechPlayer stream) per wavefront lane, one wavefront per
// workgroup.  What the reference does per sample on one CPU thread
//   frame state machine        reference src/frame.cpp:41-80
//   NaN-holding interpolation  reference src/utils.h:20-23
//   sources                    reference src/speechWaveGenerator.cpp:32-88
//   resonators                 reference src/speechWaveGenerator.cpp:90-137
//   cascade / parallel banks   reference src/speechWaveGenerator.cpp:139-182
//   mix, clip, int16           reference src/speechWaveGenerator.cpp:203-208
// each lane does here for its own utterance, with
//   * the "old" and "new" frame of the running fade staged in LDS ([param][lane],
//     conflict-free ds_read_b64), the per-sample working set in VGPRs;
//   * resonator coefficients recomputed only on samples where a fade moved them;
//   * PCM packed 8 samples per lane in registers, transposed through an XOR-swizzled
//     LDS tile and written to HBM as full 16-byte-per-lane row segments.
//
// Arithmetic is IEEE double.  This translation unit is compiled with
// -ffp-contract=off: in MODE_EXACT every multiply and add rounds separately, as in
// the reference binary; fused operations appear only where written as fma().
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace klatt {

constexpr int kNumParams = 47;
constexpr int kLanes = 64;
constexpr int kTile = 32;                 // samples per lane per output tile (64-byte row segments)
constexpr int kSlots = 45;                // parameters 1..45 live in LDS slots 0..44
constexpr int kNumRes = 14;
constexpr int kStateDoubles = 240;        // per-stream saved state (streaming path)

constexpr int MODE_EXACT = 0;
constexpr int MODE_FAST = 1;

// frame flags
constexpr uint32_t FRAME_NULL = 1u;

struct FrameMeta {           // 16 B per frame; with the 376-B parameter vector: 392 B/frame read
    uint32_t minSamples;
    uint32_t fadeSamples;    // already clamped to >= 1 (reference src/speechPlayer.cpp:36)
    int32_t userIndex;
    uint32_t flags;
};

struct UttDesc {             // 32 B per utterance
    long long frameStart;    // first frame (index into frames / meta)
    long long outStart;      // sample offset in the PCM pool, multiple of kTile
    uint32_t nFrames;
    uint32_t seed;
    uint32_t pad0, pad1;
};

struct UttResult {           // written by the kernel
    uint32_t produced;       // samples written by this launch
    uint32_t framesTaken;    // frames dequeued by this launch
    int32_t lastIndex;
    uint32_t drained;        // 1: the queue ran dry (short count)
};

struct KernelArgs {
    const double* frames;        // [nFrames][47]
    const FrameMeta* meta;       // [nFrames]
    const UttDesc* utt;          // [nUtt]
    const uint32_t* order;       // [nSlots] utterance per lane slot, 0xFFFFFFFF = empty
    int16_t* pcm;
    UttResult* result;           // [nUtt]
    double* state;               // [nUtt][kStateDoubles] or nullptr (fresh streams, nothing saved)
    const uint32_t* control;     // [nUtt] streaming only: bit0 = apply purge before synthesising
    long long nSlots;
    uint32_t maxSamples;         // per launch and utterance; 0xFFFFFFFF = until drained
    int sampleRate;
    double invSampleRate;        // RN(1/sr)
    double negPiOverSr;          // -pi/sr     (reference src/speechWaveGenerator.cpp:116)
    double twoPiOverSr;          // (2*pi)/sr  (reference src/speechWaveGenerator.cpp:118)
};

// resonator r reads frequency parameter kResF[r] and bandwidth parameter kResB[r]
// order: N0(anti), NP, c6, c5, c4, c3, c2, c1, p1..p6  (reference :149-156, :173-178)
__device__ constexpr int kResF[kNumRes] = {13, 14, 12, 11, 10, 9, 8, 7, 25, 26, 27, 28, 29, 30};
__device__ constexpr int kResB[kNumRes] = {21, 22, 20, 19, 18, 17, 16, 15, 31, 32, 33, 34, 35, 36};
// parameters the per-sample DSP reads directly (everything except 0, the f/bw pairs and 46)
__device__ constexpr int kHot[17] = {1, 2, 3, 4, 5, 6, 23, 24, 37, 38, 39, 40, 41, 42, 43, 44, 45};

// ---- arithmetic helpers -------------------------------------------------------------------

// Correctly rounded x / b from y = RN(1/b) (Markstein): 3 instructions instead of a division
// sequence.  tests/test_host_logic.py checks the identity against true division.
__device__ __forceinline__ double div_by(double x, double b, double y)
{
    double q = x * y;
    double r = __builtin_fma(-b, q, x);
    return __builtin_fma(r, y, q);
}

// fmod(x, 1) for finite x: exact, because x - trunc(x) is representable.
__device__ __forceinline__ double frac_toward_zero(double x) { return x - __builtin_trunc(x); }

// reference src/utils.h:20-23
__device__ __forceinline__ double fade_value(double from, double to, double ratio)
{
    double v = from + ((to - from) * ratio);
    return (to != to) ? from : v;
}

__device__ __forceinline__ uint32_t mix32(uint32_t x)
{
    x ^= x >> 16; x *= 0x7FEB352Du; x ^= x >> 15; x *= 0x846CA68Bu; x ^= x >> 16;
    return x;
}
// the engine's noise definition (restated by oracle/klatt_oracle.c klatt_noise31)
__device__ __forceinline__ uint32_t noise_key(uint32_t seed) { return mix32(seed ^ 0x9E3779B9u); }
__device__ __forceinline__ uint32_t noise31(uint32_t key, uint32_t k) { return mix32((k * 0x9E3779B1u) ^ key) >> 1; }

// reference src/speechWaveGenerator.cpp:112-127
template <int MODE>
__device__ __forceinline__ void resonator_coefficients(double f, double bw, bool anti, const KernelArgs& A,
                                                       double& a, double& b, double& c)
{
    double rad = exp(A.negPiOverSr * bw);
    double cc = -(rad * rad);
    double bb = rad * cos(A.twoPiOverSr * -f) * 2.0;
    double aa = 1.0 - bb - cc;
    if (anti && f != 0) {
        aa = 1.0 / aa;
        cc *= -aa;
        bb *= -aa;
    }
    a = aa; b = bb; c = cc;
}

// ---- the kernel ---------------------------------------------------------------------------

// LDS per workgroup (one wavefront):
//   oldP[kSlots][64] f64, newP[kSlots][64] f64          46,080 B
//   curFB[28][64] f64 (STREAM only: current f/bw)        14,336 B
//   tile[64 rows][kTile] i16, 16-byte chunks swizzled     4,096 B
//   rowBase[64] i64, rowCount[64] u32                       768 B
template <bool STREAM>
struct LdsLayout {
    static constexpr int kOld = 0;
    static constexpr int kNew = kOld + kSlots * kLanes * 8;
    static constexpr int kCurFB = kNew + kSlots * kLanes * 8;
    static constexpr int kTileOff = kCurFB + (STREAM ? 28 * kLanes * 8 : 0);
    static constexpr int kRowBase = kTileOff + kLanes * kTile * 2;
    static constexpr int kRowCount = kRowBase + kLanes * 8;
    static constexpr int kBytes = kRowCount + kLanes * 4;
};

template <int MODE, bool STREAM>
__global__ void __launch_bounds__(kLanes) klatt_synthesize(const KernelArgs A)
{
    using L = LdsLayout<STREAM>;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    double* const oldP = reinterpret_cast<double*>(lds + L::kOld);
    double* const newP = reinterpret_cast<double*>(lds + L::kNew);
    double* const curFB = reinterpret_cast<double*>(lds + L::kCurFB);
    uint4* const tile = reinterpret_cast<uint4*>(lds + L::kTileOff);
    long long* const rowBase = reinterpret_cast<long long*>(lds + L::kRowBase);
    uint32_t* const rowCount = reinterpret_cast<uint32_t*>(lds + L::kRowCount);

    const int lane = threadIdx.x;
    const long long slot = (long long)blockIdx.x * kLanes + lane;
    const uint32_t u = (slot < A.nSlots) ? A.order[slot] : 0xFFFFFFFFu;
    const bool live = (u != 0xFFFFFFFFu);

    UttDesc d;
    d.frameStart = 0; d.outStart = 0; d.nFrames = 0; d.seed = 0;
    if (live) d = A.utt[u];
    const double* const myFrames = A.frames + d.frameStart * kNumParams;
    const FrameMeta* const myMeta = A.meta + d.frameStart;
    const uint32_t nkey = noise_key(d.seed);

    // ---- per-lane state (fresh-handle values: reference src/frame.cpp:85-88,
    //      src/speechWaveGenerator.cpp:37,52,108-109) ----
    double cur[kNumParams];           // only cur[0] and the kHot entries are live registers
#pragma unroll
    for (int i = 0; i < kNumParams; ++i) cur[i] = 0.0;
    double old0 = 0.0, new0 = 0.0;
    double oldInc = 0.0, newInc = 0.0, invFade = 1.0;
    uint32_t cnt = 0, oldMin = 0, newMin = 0, newFade = 1;
    bool hasNew = false, oldNull = true, newNull = false;
    int32_t lastIndex = -1;
    uint32_t resMask = 0;
    uint32_t nextFrame = 0;
    uint32_t noiseIdx = 0;
    double ra[kNumRes], rb[kNumRes], rc[kNumRes], z1[kNumRes], z2[kNumRes];
#pragma unroll
    for (int r = 0; r < kNumRes; ++r) { ra[r] = 0.0; rb[r] = 2.0; rc[r] = -1.0; z1[r] = 0.0; z2[r] = 0.0; }
    double pitchPhase = 0.0, vibPhase = 0.0, aspNoise = 0.0, fricNoise = 0.0;

#pragma unroll
    for (int s = 0; s < kSlots; ++s) { oldP[s * kLanes + lane] = 0.0; newP[s * kLanes + lane] = 0.0; }
    if (STREAM) {
#pragma unroll
        for (int s = 0; s < 28; ++s) curFB[s * kLanes + lane] = 0.0;
    }

    if (STREAM && live && A.state) {
        // ---- resume a stream: see save block at the end for the layout ----
        const double* S = A.state + (size_t)u * kStateDoubles;
        if (S[239] != 0.0) {  // state valid
#pragma unroll
            for (int s = 0; s < kSlots; ++s) { oldP[s * kLanes + lane] = S[s]; newP[s * kLanes + lane] = S[45 + s]; }
#pragma unroll
            for (int s = 0; s < 28; ++s) curFB[s * kLanes + lane] = S[90 + s];
#pragma unroll
            for (int h = 0; h < 17; ++h) cur[kHot[h]] = S[118 + h];
            old0 = S[135]; new0 = S[136]; cur[0] = S[137];
#pragma unroll
            for (int r = 0; r < kNumRes; ++r) {
                ra[r] = S[138 + r]; rb[r] = S[152 + r]; rc[r] = S[166 + r]; z1[r] = S[180 + r]; z2[r] = S[194 + r];
            }
            pitchPhase = S[208]; vibPhase = S[209]; aspNoise = S[210]; fricNoise = S[211];
            oldInc = S[212]; newInc = S[213]; invFade = S[214];
            cnt = (uint32_t)S[215]; oldMin = (uint32_t)S[216]; newMin = (uint32_t)S[217]; newFade = (uint32_t)S[218];
            uint32_t fl = (uint32_t)S[219];
            hasNew = fl & 1; oldNull = fl & 2; newNull = fl & 4;
            lastIndex = (int32_t)S[220]; noiseIdx = (uint32_t)S[221]; resMask = (uint32_t)S[222];
        }
        if (A.control && (A.control[u] & 1u)) {
            // purge (reference src/frame.cpp:103-112): cut over from the current interpolated frame
            cnt = oldMin;
            if (hasNew) {
                oldNull = newNull;
                old0 = cur[0];
#pragma unroll
                for (int h = 0; h < 17; ++h) oldP[(kHot[h] - 1) * kLanes + lane] = cur[kHot[h]];
#pragma unroll
                for (int r = 0; r < kNumRes; ++r) {
                    oldP[(kResF[r] - 1) * kLanes + lane] = curFB[(2 * r) * kLanes + lane];
                    oldP[(kResB[r] - 1) * kLanes + lane] = curFB[(2 * r + 1) * kLanes + lane];
                }
                hasNew = false;
            }
        }
    }

    rowBase[lane] = d.outStart;
    rowCount[lane] = 0;
    __syncthreads();

    bool done = !live;
    bool drained = false;
    uint32_t produced = 0;
    uint32_t pk0 = 0, pk1 = 0, pk2 = 0, pk3 = 0;   // 8 packed samples
    uint32_t it = 0;                               // wave-uniform sample counter of this launch

    while (true) {
        if (!done && produced >= A.maxSamples) done = true;
        if (!__any(!done)) break;

        bool emit = false;
        if (!done) {
            // ================= frame manager, reference src/frame.cpp:41-80 =================
            cnt++;
            bool fading = false;
            if (hasNew) {
                if (cnt > newFade) {
                    // fade finished: the new request becomes the old one (:44-47)
#pragma unroll
                    for (int s = 0; s < kSlots; ++s) oldP[s * kLanes + lane] = newP[s * kLanes + lane];
                    old0 = new0; oldMin = newMin; oldInc = newInc; oldNull = newNull;
                    hasNew = false;
                } else {
                    fading = true;
                }
                emit = true;
            } else if (cnt > oldMin) {
                if (nextFrame < d.nFrames) {
                    // dequeue (:55-72)
                    const FrameMeta m = myMeta[nextFrame];
                    const double* g = myFrames + (size_t)nextFrame * kNumParams;
                    nextFrame++;
                    newMin = m.minSamples; newFade = m.fadeSamples; newNull = (m.flags & FRAME_NULL) != 0;
                    if (newNull) {
                        // silence keeps the old shape with the gain gated off (:59-63)
#pragma unroll
                        for (int s = 0; s < kSlots; ++s) newP[s * kLanes + lane] = oldP[s * kLanes + lane];
                        newP[(44 - 1) * kLanes + lane] = 0.0;
                        new0 = cur[0];
                        newInc = 0.0;
                        resMask = 0;
                    } else {
                        const double g0 = g[0];
                        const double g46 = g[46];
#pragma unroll
                        for (int s = 0; s < kSlots; ++s) newP[s * kLanes + lane] = g[s + 1];
                        new0 = g0;
                        newInc = (g46 - g0) / (double)newMin;   // reference src/frame.cpp:98
                        if (oldNull) {
                            // coming out of silence: start from the new shape, gain 0 (:64-67)
#pragma unroll
                            for (int s = 0; s < kSlots; ++s) oldP[s * kLanes + lane] = g[s + 1];
                            oldP[(44 - 1) * kLanes + lane] = 0.0;
                            old0 = g0;
                            resMask = 0;
                        } else {
                            uint32_t mk = 0;
#pragma unroll
                            for (int r = 0; r < kNumRes; ++r) {
                                const double of = oldP[(kResF[r] - 1) * kLanes + lane];
                                const double ob = oldP[(kResB[r] - 1) * kLanes + lane];
                                const bool same = (g[kResF[r]] == of) && (g[kResB[r]] == ob);
                                mk |= same ? 0u : (1u << r);
                            }
                            resMask = mk;
                        }
                    }
                    if (m.userIndex != -1) lastIndex = m.userIndex;     // :69
                    cnt = 0;                                            // :70
                    new0 += newInc * (double)newFade;                   // :71
                    invFade = 1.0 / (double)newFade;
                    hasNew = true;
                    emit = true;
                } else {
                    // queue empty: no current frame, generate() returns early (:74, wavegen :209-211)
                    done = true;
                    drained = true;
                }
            } else {
                // steady state: glide the pitch (:76-79)
                cur[0] += oldInc;
                old0 = cur[0];
                emit = true;
            }

            if (fading) {
                // interpolate (:48-53).  ratio = (double)cnt / numFadeSamples, correctly rounded
                const double ratio = div_by((double)cnt, (double)newFade, invFade);
                cur[0] = fade_value(old0, new0, ratio);
#pragma unroll
                for (int h = 0; h < 17; ++h) {
                    const int s = kHot[h] - 1;
                    cur[kHot[h]] = fade_value(oldP[s * kLanes + lane], newP[s * kLanes + lane], ratio);
                }
                // Coefficients are a pure function of (f, bw) (reference :112-127), so recomputing
                // them whenever the pair MAY have moved is equivalent to the reference's
                // recompute-on-change: on the first fade sample, and afterwards for resonators
                // whose old and new (f, bw) differ.
                const uint32_t need = (cnt == 1) ? 0x3FFFu : resMask;
#pragma unroll
                for (int r = 0; r < kNumRes; ++r) {
                    if (need & (1u << r)) {
                        const int sf = kResF[r] - 1, sb = kResB[r] - 1;
                        const double f = fade_value(oldP[sf * kLanes + lane], newP[sf * kLanes + lane], ratio);
                        const double bw = fade_value(oldP[sb * kLanes + lane], newP[sb * kLanes + lane], ratio);
                        if (STREAM) { curFB[(2 * r) * kLanes + lane] = f; curFB[(2 * r + 1) * kLanes + lane] = bw; }
                        resonator_coefficients<MODE>(f, bw, r == 0, A, ra[r], rb[r], rc[r]);
                    }
                }
            }
        }

        if (emit) {
            // ================= sources, reference src/speechWaveGenerator.cpp:72-86 =================
            double vib = 1.0;
            {
                // vibrato phase advances even when its depth is 0; fmod(0 + p, 1) == p
                const double vs = cur[2];
                if (vs != 0.0) vibPhase = frac_toward_zero(div_by(vs, (double)A.sampleRate, A.invSampleRate) + vibPhase);
                const double vo = cur[1];
                if (vo != 0.0 || vibPhase != vibPhase) vib = (sin(vibPhase * 6.283185307179586) * 0.06 * vo) + 1.0;
            }
            pitchPhase = frac_toward_zero(div_by(cur[0] * vib, (double)A.sampleRate, A.invSampleRate) + pitchPhase);
            double voice = pitchPhase;
            {
                const double un = div_by((double)noise31(nkey, noiseIdx), 2147483647.0, 0x1.00000002p-31);
                aspNoise = un + 0.75 * aspNoise;    // :40
            }
            double asp = aspNoise * 0.2;
            double turb = asp * cur[3];
            const bool glottisOpen = voice >= cur[4];
            if (!glottisOpen) turb *= 0.01;
            voice = (voice * 2.0) - 1.0;
            voice += turb;
            voice *= cur[5];
            asp *= cur[6];
            const double src = asp + voice;

            // ================= cascade, :147-158 =================
            const double x = (src * cur[44]) * 0.5;
            double o;
            {
                // anti-resonator N0: memory takes the INPUT (:133)
                const double n0 = ra[0] * x + rb[0] * z1[0] + rc[0] * z2[0];
                z2[0] = z1[0]; z1[0] = x;
                const double np = ra[1] * n0 + rb[1] * z1[1] + rc[1] * z2[1];
                z2[1] = z1[1]; z1[1] = np;
                o = fade_value(x, np, cur[23]);
            }
#pragma unroll
            for (int r = 2; r < 8; ++r) {
                const double y = ra[r] * o + rb[r] * z1[r] + rc[r] * z2[r];
                z2[r] = z1[r]; z1[r] = y;
                o = y;
            }

            // ================= frication + parallel bank, :205-206, :170-180 =================
            {
                const double un = div_by((double)noise31(nkey, noiseIdx + 1u), 2147483647.0, 0x1.00000002p-31);
                fricNoise = un + 0.75 * fricNoise;
            }
            noiseIdx += 2u;
            const double fric = fricNoise * 0.3 * cur[24];
            const double y = (fric * cur[44]) * 0.5;
            double par = 0.0;
#pragma unroll
            for (int k = 0; k < 6; ++k) {
                const int r = 8 + k;
                const double w = ra[r] * y + rb[r] * z1[r] + rc[r] * z2[r];
                z2[r] = z1[r]; z1[r] = w;
                par += (w - y) * cur[37 + k];
            }
            par = fade_value(par, y, cur[43]);

            // ================= mix, clip, quantise, :207-208 =================
            const double v = ((o + par) * cur[45]) * 4000.0;
            const double lo = (v < 32000.0) ? v : 32000.0;       // windows.h min(): NaN -> 32000
            const double cl = (lo > -32000.0) ? lo : -32000.0;
            const uint32_t s16 = (uint32_t)(int)cl & 0xFFFFu;
            produced++;
            // pack: sample (it % 8) -> half ((it%8)&1) of word (it%8)/2   (it is wave-uniform)
            const uint32_t sh = (it & 1u) * 16u;
            const uint32_t w4 = (it >> 1) & 3u;
            const uint32_t bits = s16 << sh;
            if (w4 == 0) pk0 = sh ? (pk0 | bits) : bits;
            else if (w4 == 1) pk1 = sh ? (pk1 | bits) : bits;
            else if (w4 == 2) pk2 = sh ? (pk2 | bits) : bits;
            else pk3 = sh ? (pk3 | bits) : bits;
        }

        // ---- output staging: every 8th sample a 16-byte chunk goes to the LDS tile; every
        //      kTile samples the tile is written out as 16 B per lane, whole row segments ----
        if ((it & 7u) == 7u) {
            const uint32_t chunk = (it % kTile) >> 3;
            tile[lane * (kTile / 8) + (chunk ^ (lane & (kTile / 8 - 1)))] = make_uint4(pk0, pk1, pk2, pk3);
        }
        it++;
        if ((it % kTile) == 0) {
            rowCount[lane] = produced;
            __syncthreads();
            const uint32_t tileStart = it - kTile;
            constexpr int kChunksPerRow = kTile / 8;
            constexpr int kRowsPerPass = kLanes / kChunksPerRow;
#pragma unroll
            for (int p = 0; p < kLanes / kRowsPerPass; ++p) {
                const int row = p * kRowsPerPass + lane / kChunksPerRow;
                const int chunk = lane % kChunksPerRow;
                const uint4 val = tile[row * kChunksPerRow + (chunk ^ (row & (kChunksPerRow - 1)))];
                if (rowCount[row] > tileStart) {
                    uint4* dst = reinterpret_cast<uint4*>(A.pcm + rowBase[row] + tileStart + chunk * 8);
                    *dst = val;
                }
            }
            __syncthreads();
        }
    }

    // ---- flush the last, partial tile ----
    if ((it % kTile) != 0) {
        const uint32_t tileStart = it - (it % kTile);
        if ((it & 7u) != 0u) {
            const uint32_t chunk = (it % kTile) >> 3;
            tile[lane * (kTile / 8) + (chunk ^ (lane & (kTile / 8 - 1)))] = make_uint4(pk0, pk1, pk2, pk3);
        }
        rowCount[lane] = produced;
        __syncthreads();
        constexpr int kChunksPerRow = kTile / 8;
        constexpr int kRowsPerPass = kLanes / kChunksPerRow;
#pragma unroll
        for (int p = 0; p < kLanes / kRowsPerPass; ++p) {
            const int row = p * kRowsPerPass + lane / kChunksPerRow;
            const int chunk = lane % kChunksPerRow;
            const uint4 val = tile[row * kChunksPerRow + (chunk ^ (row & (kChunksPerRow - 1)))];
            if (rowCount[row] > tileStart + (uint32_t)chunk * 8u) {
                uint4* dst = reinterpret_cast<uint4*>(A.pcm + rowBase[row] + tileStart + chunk * 8);
                *dst = val;
            }
        }
    }

    if (live) {
        UttResult res;
        res.produced = produced; res.framesTaken = nextFrame; res.lastIndex = lastIndex; res.drained = drained ? 1u : 0u;
        A.result[u] = res;
    }

    if (STREAM && live && A.state) {
        double* S = A.state + (size_t)u * kStateDoubles;
#pragma unroll
        for (int s = 0; s < kSlots; ++s) { S[s] = oldP[s * kLanes + lane]; S[45 + s] = newP[s * kLanes + lane]; }
#pragma unroll
        for (int s = 0; s < 28; ++s) S[90 + s] = curFB[s * kLanes + lane];
#pragma unroll
        for (int h = 0; h < 17; ++h) S[118 + h] = cur[kHot[h]];
        S[135] = old0; S[136] = new0; S[137] = cur[0];
#pragma unroll
        for (int r = 0; r < kNumRes; ++r) {
            S[138 + r] = ra[r]; S[152 + r] = rb[r]; S[166 + r] = rc[r]; S[180 + r] = z1[r]; S[194 + r] = z2[r];
        }
        S[208] = pitchPhase; S[209] = vibPhase; S[210] = aspNoise; S[211] = fricNoise;
        S[212] = oldInc; S[213] = newInc; S[214] = invFade;
        S[215] = (double)cnt; S[216] = (double)oldMin; S[217] = (double)newMin; S[218] = (double)newFade;
        S[219] = (double)((hasNew ? 1u : 0u) | (oldNull ? 2u : 0u) | (newNull ? 4u : 0u));
        S[220] = (double)lastIndex; S[221] = (double)noiseIdx; S[222] = (double)resMask;
        S[239] = 1.0;
    }
}

}  // namespace klatt
