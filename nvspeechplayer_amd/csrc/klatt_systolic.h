// klatt_systolic.h -- stage-parallel batch kernel: one workgroup = 4 wavefronts = 4 pipeline stages
// over the SAME 64 utterances (lane = utterance in every wave).
//
// The lane-per-utterance kernel (klatt_device.h) is bound by f64 VALU issue of ONE wave per 64
// utterances and by its register/LDS footprint (3 waves per CU).  The per-sample work of the
// reference's generate() loop (reference src/speechWaveGenerator.cpp:197-214) is a chain
//     source -> N0 -> NP -> r6..r1 -> (+ parallel bank) -> gain/clip
// whose links only pass one double per sample.  Here the links are cut into four stages that run
// on the four SIMDs of a CU at the same time, each on a different 8-sample chunk, handing
// chunks over through double-buffered LDS pipes with one workgroup barrier per chunk:
//
//   noisy launch   S0  frame(0,1-6,44) + glottal source + aspiration noise        -> x
//                  S1  N0, NP (caNP mix), r6, r5, r4                               x -> o
//                  S3  frication noise, parallel r1..r4 (partial sum)              -> (y, part)
//                  S2  r3, r2, r1; parallel r5, r6, bypass mix; gain, clip, int16  o,(y,part) -> PCM
//   quiet launch   S0  frame + glottal source                                     -> x
//                  S1  N0, NP, r6                                                  x -> o
//                  S2  r5, r4, r3                                                  o -> o
//                  S3  r2, r1; gain, clip, int16                                   o -> PCM
//
// Every stage runs its own copy of the frame state machine (reference src/frame.cpp:41-80) for
// just the parameters it needs, entirely in registers, so no operation, operand or rounding
// differs from the lane kernel / the reference: the stages compute the same values in the same
// order, only on different SIMDs.  Each wave therefore carries a quarter of the state
// (no AGPR traffic), LDS holds only the pipes and the PCM tile, and a 4096-utterance batch
// becomes 256 wavefronts instead of 64.
#pragma once

#include "klatt_device.h"

namespace klatt {

#ifndef KLATT_CHUNK
#define KLATT_CHUNK 8
#endif
#ifndef KLATT_UNROLL
#define KLATT_UNROLL 8
#endif
#ifndef KLATT_MINWAVES
#define KLATT_MINWAVES 1
#endif
constexpr int kChunk = KLATT_CHUNK;           // samples per pipeline hand-over
constexpr int kStages = 4;
#define KLATT_STR2(x) #x
#define KLATT_STR(x) KLATT_STR2(x)

// LDS: four pipes [2 buffers][kChunk][64 lanes] f64, the PCM tile (last stage only), row info
struct SysLds {
    static constexpr int kPipeBytes = 2 * kChunk * kLanes * 8;       // 8 KB at kChunk = 8
    static constexpr int kPipe0 = 0;
    static constexpr int kTileOff = 4 * kPipeBytes;
    static constexpr int kRowBase = kTileOff + kLanes * kTileStride;
    static constexpr int kRowCount = kRowBase + kLanes * 8;
    static constexpr int kMaxLen = kRowCount + kLanes * 4;
    static constexpr int kBytes = kMaxLen + 16;
};

// ---- the frame state machine, restricted to a stage's parameter subset ----------------------
// NPARAM tracked parameters P[0..NPARAM), resonators r use (P[RF[r]], P[RB[r]]).
template <int NPARAM, int NRES>
struct StageFrame {
    double old[NPARAM > 0 ? NPARAM : 1], nw[NPARAM > 0 ? NPARAM : 1], cur[NPARAM > 0 ? NPARAM : 1];
    double ra[NRES > 0 ? NRES : 1], rb[NRES > 0 ? NRES : 1], rc[NRES > 0 ? NRES : 1];
    double z1[NRES > 0 ? NRES : 1], z2[NRES > 0 ? NRES : 1];
    double invFade;
    uint32_t cnt, oldMin, newMin, newFade, nextFrame, resMask, parMask, produced;
    bool hasNew, oldNull, newNull, done;
};

// pitch (parameter 0) needs the glide state; only the source stage has it
struct PitchState {
    double cur0, old0, new0, oldInc, newInc;
};

template <class SF>
__device__ __forceinline__ void stage_frame_init(SF& f, bool live)
{
#pragma unroll
    for (int i = 0; i < (int)(sizeof(f.old) / 8); ++i) { f.old[i] = 0.0; f.nw[i] = 0.0; f.cur[i] = 0.0; }
#pragma unroll
    for (int r = 0; r < (int)(sizeof(f.ra) / 8); ++r) { f.ra[r] = 0.0; f.rb[r] = 2.0; f.rc[r] = -1.0; f.z1[r] = 0.0; f.z2[r] = 0.0; }
    f.invFade = 1.0;
    f.cnt = 0; f.oldMin = 0; f.newMin = 0; f.newFade = 1; f.nextFrame = 0; f.resMask = 0; f.parMask = 0; f.produced = 0;
    f.hasNew = false; f.oldNull = true; f.newNull = false; f.done = !live;
}

// Stage descriptor: which parameters, which of them form resonators, which one is preFormantGain.
// GAIN = index into P of parameter 44 (or -1): NULL frames force it to 0 (reference src/frame.cpp:61,66).
template <int NPARAM_, int NRES_, int GAIN_, bool PITCH_>
struct StageDesc {
    static constexpr int NPARAM = NPARAM_, NRES = NRES_, GAIN = GAIN_;
    static constexpr bool PITCH = PITCH_;
};

// one event sample (fade end / dequeue / end of queue) for a stage; mirrors event_step() of klatt_device.h
template <class D, class SF>
__device__ __forceinline__ bool stage_event(SF& f, PitchState* ps, int32_t* lastIndex, const int* P,
                                            const int* RF, const int* RB,
                                            const UttDesc& d, const double* myFrames, const FrameMeta* myMeta)
{
    if (f.hasNew) {   // fade finished (reference src/frame.cpp:44-47)
#pragma unroll
        for (int k = 0; k < D::NPARAM; ++k) f.old[k] = f.nw[k];
        f.oldMin = f.newMin; f.oldNull = f.newNull;
        if (D::PITCH) { ps->old0 = ps->new0; ps->oldInc = ps->newInc; }
        f.hasNew = false;
        return true;
    }
    if (f.nextFrame >= d.nFrames) { f.done = true; return false; }   // queue empty (:74)
    const FrameMeta m = myMeta[f.nextFrame];
    const double* g = myFrames + (size_t)f.nextFrame * kNumParams;
    f.nextFrame++;
    f.newMin = m.minSamples; f.newFade = m.fadeSamples; f.newNull = (m.flags & FRAME_NULL) != 0;
    if (f.newNull) {   // (:59-63)
#pragma unroll
        for (int k = 0; k < D::NPARAM; ++k) f.nw[k] = f.old[k];
        if (D::GAIN >= 0) f.nw[D::GAIN >= 0 ? D::GAIN : 0] = 0.0;
        if (D::PITCH) { ps->new0 = ps->cur0; ps->newInc = 0.0; }
        f.resMask = 0;
        f.parMask = (D::GAIN >= 0 && f.old[D::GAIN >= 0 ? D::GAIN : 0] != 0.0) ? (1u << (D::GAIN >= 0 ? D::GAIN : 0)) : 0u;
    } else {
#pragma unroll
        for (int k = 0; k < D::NPARAM; ++k) f.nw[k] = g[P[k]];
        if (D::PITCH) {
            const double g0 = g[0], g46 = g[46];
            ps->new0 = g0;
            ps->newInc = (g46 - g0) / (double)f.newMin;   // reference src/frame.cpp:98
            if (f.oldNull) ps->old0 = g0;
        }
        if (f.oldNull) {   // (:64-67)
#pragma unroll
            for (int k = 0; k < D::NPARAM; ++k) f.old[k] = f.nw[k];
            if (D::GAIN >= 0) f.old[D::GAIN >= 0 ? D::GAIN : 0] = 0.0;
            f.resMask = 0;
            f.parMask = (D::GAIN >= 0 && f.nw[D::GAIN >= 0 ? D::GAIN : 0] != 0.0) ? (1u << (D::GAIN >= 0 ? D::GAIN : 0)) : 0u;
        } else {
            uint32_t mk = 0;
#pragma unroll
            for (int r = 0; r < D::NRES; ++r) {
                const bool same = (f.nw[RF[r]] == f.old[RF[r]]) && (f.nw[RB[r]] == f.old[RB[r]]);
                mk |= same ? 0u : (1u << r);
            }
            f.resMask = mk;
            uint32_t pm = 0;
#pragma unroll
            for (int k = 0; k < D::NPARAM; ++k) pm |= (f.nw[k] == f.old[k]) ? 0u : (1u << k);   // NaN (hold) counts as moving: harmless
            f.parMask = pm;
        }
    }
    if (lastIndex && m.userIndex != -1) *lastIndex = m.userIndex;   // (:69)
    f.cnt = 0;                                                       // (:70)
    if (D::PITCH) ps->new0 += ps->newInc * (double)f.newFade;        // (:71)
    f.invFade = 1.0 / (double)f.newFade;
    f.hasNew = true;
    return true;
}

// OR of a per-lane mask over the wavefront, as a wave-uniform (scalar) value
__device__ __forceinline__ uint32_t wave_or(uint32_t m)
{
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) m |= (uint32_t)__shfl_xor((int)m, o);
    return (uint32_t)__builtin_amdgcn_readfirstlane((int)m);
}

// Recompute the coefficients of the resonators named in `need` (per lane) / `waveNeed` (OR over the wave,
// wave-uniform).  One rolled loop with a single inlined exp/cos body: r is scalar, so picking the
// resonator's (f, bw) and storing its (a, b, c) are scalar selects.  Lanes outside `need` keep theirs.
template <int MODE, int NRES, bool ANTI0, class SF>
__device__ __forceinline__ void update_coefficients(SF& f, const KernelArgs& A, const int* RF, const int* RB,
                                                    uint32_t need, uint32_t waveNeed)
{
    if (NRES == 0) return;
#pragma nounroll
    for (int r = 0; r < NRES; ++r) {
        if (!((waveNeed >> r) & 1u)) continue;
        double fq = 0.0, bq = 0.0;
#pragma unroll
        for (int q = 0; q < NRES; ++q) { fq = (q == r) ? f.cur[RF[q]] : fq; bq = (q == r) ? f.cur[RB[q]] : bq; }
        const Coef k = resonator_coefficients_inline<MODE>(fq, bq, ANTI0 && r == 0, A.negPiOverSr, A.twoPiOverSr);
        const bool mine = (need >> r) & 1u;
#pragma unroll
        for (int q = 0; q < NRES; ++q) {
            const bool hit = mine && (q == r);
            f.ra[q] = hit ? k.a : f.ra[q]; f.rb[q] = hit ? k.b : f.rb[q]; f.rc[q] = hit ? k.c : f.rc[q];
        }
    }
}

// one fade sample's update for a stage (reference src/frame.cpp:48-53); ANTI0: resonator 0 is the anti-resonator
template <class D, int MODE, bool ANTI0, class SF>
__device__ __forceinline__ void stage_fade(SF& f, PitchState* ps, const KernelArgs& A,
                                           const int* RF, const int* RB)
{
    const double ratio = div_by((double)f.cnt, (double)f.newFade, f.invFade);
    if (D::PITCH) ps->cur0 = fade_value(ps->old0, ps->new0, ratio);
#pragma unroll
    for (int k = 0; k < D::NPARAM; ++k) f.cur[k] = fade_value(f.old[k], f.nw[k], ratio);
    const uint32_t need = (f.cnt == 1) ? 0x3FFFu : f.resMask;
    uint32_t waveNeed = 0;   // OR over the lanes that are executing this (ballots see only active lanes)
#pragma unroll
    for (int r = 0; r < D::NRES; ++r) waveNeed |= __any(need & (1u << r)) ? (1u << r) : 0u;
    update_coefficients<MODE, D::NRES, ANTI0>(f, A, RF, RB, need, waveNeed);
}

// A fade sample inside a stretch where EVERY live lane is past its first fade sample: only parameters
// (and resonators) that move in SOME lane are touched.  For a lane whose own old == new the update
// recomputes the value it already holds (old + 0*ratio; coefficients are a pure function), so the result
// is the one stage_fade() would give.  wPar / wRes are wave-uniform, the branches are scalar.
template <class D, int MODE, bool ANTI0, class SF>
__device__ __forceinline__ void stage_fade_masked(SF& f, PitchState* ps, const KernelArgs& A, const int* RF, const int* RB,
                                                  uint32_t wPar, uint32_t wRes)
{
    f.cnt++;
    const double ratio = div_by((double)f.cnt, (double)f.newFade, f.invFade);
    if (D::PITCH) ps->cur0 = fade_value(ps->old0, ps->new0, ratio);
#pragma unroll
    for (int k = 0; k < D::NPARAM; ++k)
        if (wPar & (1u << k)) f.cur[k] = fade_value(f.old[k], f.nw[k], ratio);
    update_coefficients<MODE, D::NRES, ANTI0>(f, A, RF, RB, wRes, wRes);
}

// advance the state machine by one sample; returns true when a sample is emitted
template <class D, int MODE, bool ANTI0, class SF>
__device__ __forceinline__ bool stage_advance(SF& f, PitchState* ps, int32_t* lastIndex, const KernelArgs& A,
                                              const int* P,
                                              const int* RF, const int* RB,
                                              const UttDesc& d, const double* myFrames, const FrameMeta* myMeta)
{
    if (f.done) return false;
    f.cnt++;
    if (f.hasNew && f.cnt <= f.newFade) {
        stage_fade<D, MODE, ANTI0>(f, ps, A, RF, RB);
        return true;
    }
    if (!f.hasNew && f.cnt <= f.oldMin) {
        if (D::PITCH) { ps->cur0 += ps->oldInc; ps->old0 = ps->cur0; }   // glide (reference src/frame.cpp:76-79)
        return true;
    }
    return stage_event<D>(f, ps, lastIndex, P, RF, RB, d, myFrames, myMeta);
}

template <int MODE>
__device__ __forceinline__ double resonate(double& z1, double& z2, double a, double b, double c, double in)
{
    const double y = dot3<MODE>(a, in, b, z1, c, z2);
    z2 = z1; z1 = y;
    return y;
}

// all live lanes steady (0) / all fading beyond their first fade sample (1), with at least kChunk samples left; else -1
template <class SF>
__device__ __forceinline__ int chunk_kind(const SF& f)
{
    const uint32_t rem = f.hasNew ? (f.newFade - f.cnt) : (f.oldMin > f.cnt ? f.oldMin - f.cnt : 0u);
    const bool roomy = f.done || rem >= (uint32_t)kChunk;
    if (!__all(roomy)) return -1;
    if (!__any(!f.done && f.hasNew)) return 0;
    if (!__any(!f.done && (!f.hasNew || f.cnt == 0))) return 1;   // all fading and past the first fade sample
    return -1;
}

// ---- the kernel ---------------------------------------------------------------------------
template <int MODE, bool NOISE>
__global__ void __launch_bounds__(kLanes * kStages, KLATT_MINWAVES) klatt_systolic(const KernelArgs A)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    double* const pipeX = reinterpret_cast<double*>(lds + SysLds::kPipe0);                          // S0 -> S1
    double* const pipeO = reinterpret_cast<double*>(lds + SysLds::kPipe0 + SysLds::kPipeBytes);     // S1 -> S2
    double* const pipeA = reinterpret_cast<double*>(lds + SysLds::kPipe0 + 2 * SysLds::kPipeBytes); // noisy: y   | quiet: S2 -> S3
    double* const pipeB = reinterpret_cast<double*>(lds + SysLds::kPipe0 + 3 * SysLds::kPipeBytes); // noisy: partial sum
    unsigned char* const tile = lds + SysLds::kTileOff;
    long long* const rowBase = reinterpret_cast<long long*>(lds + SysLds::kRowBase);
    uint32_t* const rowCount = reinterpret_cast<uint32_t*>(lds + SysLds::kRowCount);
    uint32_t* const maxLenP = reinterpret_cast<uint32_t*>(lds + SysLds::kMaxLen);

    const int lane = threadIdx.x & (kLanes - 1);
    const int stage = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const long long slot = (long long)blockIdx.x * kLanes + lane;
    const uint32_t u = (slot < A.nSlots) ? A.order[slot] : 0xFFFFFFFFu;
    const bool live = (u != 0xFFFFFFFFu);

    UttDesc d;
    d.frameStart = 0; d.outStart = 0; d.nFrames = 0; d.seed = 0; d.flags = 0; d.length = 0;
    if (live) d = A.utt[u];
    const double* const myFrames = A.frames + d.frameStart * kNumParams;
    const FrameMeta* const myMeta = A.meta + d.frameStart;
    const uint32_t nkey = noise_key(d.seed);

    if (threadIdx.x == 0) *maxLenP = 0;
    __syncthreads();
    if (stage == 0) atomicMax(maxLenP, d.length);
    if (stage == (NOISE ? 2 : 3)) { rowBase[lane] = d.outStart; rowCount[lane] = 0; }
    __syncthreads();
    const uint32_t maxLen = *maxLenP;
    const int nChunks = (int)((maxLen + kChunk - 1) / kChunk);
    const int nIter = nChunks + (NOISE ? 2 : 3);   // the final stage lags 2 (noisy) or 3 (quiet) chunks; same trip count in every wave

    // pipe slot of sample i of chunk c
#define PIPE(p, c, i) (p)[(((c) & 1) * kChunk + (i)) * kLanes + lane]
#ifdef KLATT_STAMPS
    unsigned long long stWork = 0, stWait = 0, stT0 = 0, stT1 = 0, stFastN = 0, stGenN = 0, stFadeN = 0, stFastC = 0, stFadeC = 0, stGenC = 0;
    int stKind = 2;
#define STAMP_BEGIN() stT0 = __builtin_amdgcn_s_memtime()
#define STAMP_WORKED() do { stT1 = __builtin_amdgcn_s_memtime(); stWork += stT1 - stT0; if (stKind == 0) stFastC += stT1 - stT0; else if (stKind == 1) stFadeC += stT1 - stT0; else if (stKind == -1) stGenC += stT1 - stT0; stKind = 2; } while (0)
#define STAMP_SYNCED() do { stWait += __builtin_amdgcn_s_memtime() - stT1; } while (0)
#define STAMP_KIND(k) do { stKind = (k); if ((k) == 0) stFastN++; else if ((k) == 1) stFadeN++; else stGenN++; } while (0)
#else
#define STAMP_KIND(k)
#define STAMP_BEGIN()
#define STAMP_WORKED()
#define STAMP_SYNCED()
#endif

    if (stage == 0) {
        // ================= S0: frame + glottal source (+ aspiration noise) =================
        // tracked: 1 vibratoPitchOffset, 2 vibratoSpeed, 3 turbulence, 4 openQuotient, 5 voiceAmplitude,
        //          6 aspirationAmplitude, 44 preFormantGain (quiet launches never read 3, 4, 6)
        using D = StageDesc<7, 0, 6, true>;
        constexpr int P[7] = {1, 2, 3, 4, 5, 6, 44};
        constexpr int RF[1] = {0}, RB[1] = {0};
        StageFrame<7, 0> f;
        PitchState ps;
        stage_frame_init(f, live);
        ps.cur0 = 0.0; ps.old0 = 0.0; ps.new0 = 0.0; ps.oldInc = 0.0; ps.newInc = 0.0;
        double pitchPhase = 0.0, vibPhase = 0.0, aspNoise = 0.0;
        uint32_t noiseIdx = 0;
        int32_t lastIndex = -1;
        bool vibFrames = false;

        auto source = [&](bool waveVib) __attribute__((always_inline)) -> double {
            double vib = 1.0;
            if (waveVib) {
                const double vs = f.cur[1];
                const double adv = frac_toward_zero(div_by(vs, A.sampleRateF, A.invSampleRate) + vibPhase);
                vibPhase = (vs != 0.0) ? adv : vibPhase;
                vib = (sin(vibPhase * 6.283185307179586) * 0.06 * f.cur[0]) + 1.0;
            }
            pitchPhase = frac_toward_zero(div_by(ps.cur0 * vib, A.sampleRateF, A.invSampleRate) + pitchPhase);
            double voice = (pitchPhase * 2.0) - 1.0;
            double src;
            if (NOISE) {
                aspNoise = noise_uniform(nkey, noiseIdx) + 0.75 * aspNoise;
                noiseIdx += 2u;
                double asp = aspNoise * 0.2;
                double turb = asp * f.cur[2];
                turb = (pitchPhase >= f.cur[3]) ? turb : turb * 0.01;
                voice += turb;
                voice *= f.cur[4];
                asp *= f.cur[5];
                src = asp + voice;
            } else {
                src = voice * f.cur[4];
            }
            return (src * f.cur[6]) * 0.5;
        };
        auto vib_live_now = [&]() __attribute__((always_inline)) -> bool {
            return vibFrames || f.cur[0] != 0.0 || f.cur[1] != 0.0 || vibPhase != vibPhase;
        };

        for (int iter = 0; iter < nIter; ++iter) {
            STAMP_BEGIN();
            const int c = iter;
            if (c < nChunks) {
                const int kind = __any(!f.done && vib_live_now()) ? -1 : chunk_kind(f);
                STAMP_KIND(kind);
                if (kind == 0) {
                    if (!f.done) {
_Pragma(KLATT_STR(unroll KLATT_UNROLL))
                        for (int i = 0; i < kChunk; ++i) {
                            ps.cur0 += ps.oldInc;
                            PIPE(pipeX, c, i) = source(false);
                        }
                        ps.old0 = ps.cur0;
                        f.cnt += kChunk; f.produced += kChunk;
                    }
                } else if (kind == 1) {
                    const uint32_t wPar = wave_or(f.parMask);
                    if (!f.done) {
#pragma nounroll
                        for (int i = 0; i < kChunk; ++i) {
                            stage_fade_masked<D, MODE, false>(f, &ps, A, RF, RB, wPar, 0u);
                            PIPE(pipeX, c, i) = source(false);
                        }
                        f.produced += kChunk;
                    }
                } else {
#pragma nounroll
                    for (int i = 0; i < kChunk; ++i) {
                        const bool wasNew = f.hasNew;
                        const bool emit = stage_advance<D, MODE, false>(f, &ps, &lastIndex, A, P, RF, RB, d, myFrames, myMeta);
                        if (emit && f.hasNew && !wasNew)
                            vibFrames = f.old[0] != 0.0 || f.old[1] != 0.0 || f.nw[0] != 0.0 || f.nw[1] != 0.0;
                        const bool waveVib = __any(emit && vib_live_now());
                        if (emit) { PIPE(pipeX, c, i) = source(waveVib); f.produced++; }
                    }
                }
            }
            STAMP_WORKED();
            __syncthreads();
            STAMP_SYNCED();
        }
        if (live) {
            UttResult res;
            res.produced = f.produced; res.framesTaken = f.nextFrame; res.lastIndex = lastIndex; res.drained = 1u;
            A.result[u] = res;
        }
    } else if (stage == 1) {
        // ================= S1: N0 (anti), NP mixed by caNP, r6 [, r5, r4] =================
        constexpr int NR = NOISE ? 5 : 3;
        using D = StageDesc<2 * NR + 1, NR, -1, false>;
        // parameter list: (f, bw) of N0, NP, r6 [, r5, r4], then caNP
        constexpr int P[11] = {13, 21, 14, 22, 12, 20, NOISE ? 11 : 23, 19, 10, 18, 23};
        constexpr int RF[5] = {0, 2, 4, 6, 8};
        constexpr int RB[5] = {1, 3, 5, 7, 9};
        constexpr int CANP = 2 * NR;
        StageFrame<2 * NR + 1, NR> f;
        stage_frame_init(f, live);
        auto dsp = [&](double x) __attribute__((always_inline)) -> double {
            const double n0 = dot3<MODE>(f.ra[0], x, f.rb[0], f.z1[0], f.rc[0], f.z2[0]);
            f.z2[0] = f.z1[0]; f.z1[0] = x;                       // anti-resonator remembers its INPUT (:133)
            const double np = resonate<MODE>(f.z1[1], f.z2[1], f.ra[1], f.rb[1], f.rc[1], n0);
            double o = fade_value(x, np, f.cur[CANP]);
#pragma unroll
            for (int r = 2; r < NR; ++r) o = resonate<MODE>(f.z1[r], f.z2[r], f.ra[r], f.rb[r], f.rc[r], o);
            return o;
        };
        for (int iter = 0; iter < nIter; ++iter) {
            STAMP_BEGIN();
            const int c = iter - 1;
            if (c >= 0 && c < nChunks) {
                const int kind = chunk_kind(f);
                STAMP_KIND(kind);
                if (kind == 0) {
                    if (!f.done) {
_Pragma(KLATT_STR(unroll KLATT_UNROLL))
                        for (int i = 0; i < kChunk; ++i) PIPE(pipeO, c, i) = dsp(PIPE(pipeX, c, i));
                        f.cnt += kChunk;
                    }
                } else if (kind == 1) {
                    const uint32_t wPar = wave_or(f.parMask), wRes = wave_or(f.resMask);
                    if (!f.done) {
#pragma nounroll
                        for (int i = 0; i < kChunk; ++i) {
                            stage_fade_masked<D, MODE, true>(f, nullptr, A, RF, RB, wPar, wRes);
                            PIPE(pipeO, c, i) = dsp(PIPE(pipeX, c, i));
                        }
                    }
                } else {
#pragma nounroll
                    for (int i = 0; i < kChunk; ++i) {
                        const bool emit = stage_advance<D, MODE, true>(f, nullptr, nullptr, A, P, RF, RB, d, myFrames, myMeta);
                        if (emit) PIPE(pipeO, c, i) = dsp(PIPE(pipeX, c, i));
                    }
                }
            }
            STAMP_WORKED();
            __syncthreads();
            STAMP_SYNCED();
        }
    } else if (NOISE && stage == 3) {
        // ================= noisy S3: frication noise, parallel r1..r4 partial sum =================
        // tracked: (pf, pb) of parallel 1..4, then 24 fricationAmplitude, 44 preFormantGain, pa1..4 (37..40)
        using D = StageDesc<14, 4, 9, false>;
        constexpr int P[14] = {25, 31, 26, 32, 27, 33, 28, 34, 24, 44, 37, 38, 39, 40};
        constexpr int RF[4] = {0, 2, 4, 6}, RB[4] = {1, 3, 5, 7};
        StageFrame<14, 4> f;
        stage_frame_init(f, live);
        double fricNoise = 0.0;
        uint32_t noiseIdx = 1;
        auto dsp = [&](double& yOut) __attribute__((always_inline)) -> double {
            fricNoise = noise_uniform(nkey, noiseIdx) + 0.75 * fricNoise;
            noiseIdx += 2u;
            const double fric = fricNoise * 0.3 * f.cur[8];
            const double y = (fric * f.cur[9]) * 0.5;
            double par = 0.0;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const double w = resonate<MODE>(f.z1[r], f.z2[r], f.ra[r], f.rb[r], f.rc[r], y);
                par += (w - y) * f.cur[10 + r];
            }
            yOut = y;
            return par;
        };
        for (int iter = 0; iter < nIter; ++iter) {
            STAMP_BEGIN();
            const int c = iter - 1;
            if (c >= 0 && c < nChunks) {
                const int kind = chunk_kind(f);
                STAMP_KIND(kind);
                if (kind == 0) {
                    if (!f.done) {
_Pragma(KLATT_STR(unroll KLATT_UNROLL))
                        for (int i = 0; i < kChunk; ++i) { double y; const double p = dsp(y); PIPE(pipeA, c, i) = y; PIPE(pipeB, c, i) = p; }
                        f.cnt += kChunk;
                    }
                } else if (kind == 1) {
                    const uint32_t wPar = wave_or(f.parMask), wRes = wave_or(f.resMask);
                    if (!f.done) {
#pragma nounroll
                        for (int i = 0; i < kChunk; ++i) {
                            stage_fade_masked<D, MODE, false>(f, nullptr, A, RF, RB, wPar, wRes);
                            double y; const double p = dsp(y); PIPE(pipeA, c, i) = y; PIPE(pipeB, c, i) = p;
                        }
                    }
                } else {
#pragma nounroll
                    for (int i = 0; i < kChunk; ++i) {
                        const bool emit = stage_advance<D, MODE, false>(f, nullptr, nullptr, A, P, RF, RB, d, myFrames, myMeta);
                        if (emit) { double y; const double p = dsp(y); PIPE(pipeA, c, i) = y; PIPE(pipeB, c, i) = p; }
                    }
                }
            }
            STAMP_WORKED();
            __syncthreads();
            STAMP_SYNCED();
        }
    } else if (!NOISE && stage == 2) {
        // ================= quiet S2: r5, r4, r3 =================
        using D = StageDesc<6, 3, -1, false>;
        constexpr int P[6] = {11, 19, 10, 18, 9, 17};
        constexpr int RF[3] = {0, 2, 4}, RB[3] = {1, 3, 5};
        StageFrame<6, 3> f;
        stage_frame_init(f, live);
        auto dsp = [&](double o) __attribute__((always_inline)) -> double {
#pragma unroll
            for (int r = 0; r < 3; ++r) o = resonate<MODE>(f.z1[r], f.z2[r], f.ra[r], f.rb[r], f.rc[r], o);
            return o;
        };
        for (int iter = 0; iter < nIter; ++iter) {
            STAMP_BEGIN();
            const int c = iter - 2;
            if (c >= 0 && c < nChunks) {
                const int kind = chunk_kind(f);
                STAMP_KIND(kind);
                if (kind == 0) {
                    if (!f.done) {
_Pragma(KLATT_STR(unroll KLATT_UNROLL))
                        for (int i = 0; i < kChunk; ++i) PIPE(pipeA, c, i) = dsp(PIPE(pipeO, c, i));
                        f.cnt += kChunk;
                    }
                } else if (kind == 1) {
                    const uint32_t wPar = wave_or(f.parMask), wRes = wave_or(f.resMask);
                    if (!f.done) {
#pragma nounroll
                        for (int i = 0; i < kChunk; ++i) {
                            stage_fade_masked<D, MODE, false>(f, nullptr, A, RF, RB, wPar, wRes);
                            PIPE(pipeA, c, i) = dsp(PIPE(pipeO, c, i));
                        }
                    }
                } else {
#pragma nounroll
                    for (int i = 0; i < kChunk; ++i) {
                        const bool emit = stage_advance<D, MODE, false>(f, nullptr, nullptr, A, P, RF, RB, d, myFrames, myMeta);
                        if (emit) PIPE(pipeA, c, i) = dsp(PIPE(pipeO, c, i));
                    }
                }
            }
            STAMP_WORKED();
            __syncthreads();
            STAMP_SYNCED();
        }
    } else {
        // ================= final stage: rest of the cascade, (parallel r4..r6 + bypass), gain, clip, PCM ===
        // noisy (stage 2): r3, r2, r1 | parallel 5, 6 | pa5, pa6, parallelBypass, outputGain
        // quiet (stage 3): r2, r1 | outputGain
        constexpr int NC = NOISE ? 3 : 2;                 // cascade resonators here
        constexpr int NR = NOISE ? 5 : 2;
        constexpr int NPAR = NOISE ? 14 : 5;
        using D = StageDesc<NPAR, NR, -1, false>;
        constexpr int P[14] = {NOISE ? 9 : 8, NOISE ? 17 : 16, NOISE ? 8 : 7, NOISE ? 16 : 15, NOISE ? 7 : 45, 15,
                               29, 35, 30, 36, 41, 42, 43, 45};
        constexpr int RF[5] = {0, 2, NOISE ? 4 : 0, 6, 8};
        constexpr int RB[5] = {1, 3, NOISE ? 5 : 0, 7, 9};
        constexpr int OUTGAIN = NOISE ? 13 : 4;
        StageFrame<NPAR, NR> f;
        stage_frame_init(f, live);
        int16_t* const myRow = reinterpret_cast<int16_t*>(tile + lane * kTileStride);
        const int lag = NOISE ? 2 : 3;

        auto finish = [&](double o, double y, double part) __attribute__((always_inline)) -> uint32_t {
#pragma unroll
            for (int r = 0; r < NC; ++r) o = resonate<MODE>(f.z1[r], f.z2[r], f.ra[r], f.rb[r], f.rc[r], o);
            double mix = o;
            if (NOISE) {
                double par = part;
#pragma unroll
                for (int r = 3; r < 5; ++r) {
                    const double w = resonate<MODE>(f.z1[r], f.z2[r], f.ra[r], f.rb[r], f.rc[r], y);
                    par += (w - y) * f.cur[10 + (r - 3)];
                }
                par = fade_value(par, y, f.cur[12]);
                mix = o + par;
            }
            const double v = (mix * f.cur[OUTGAIN]) * 4000.0;
            const double lo = (v < 32000.0) ? v : 32000.0;
            const double cl = (lo > -32000.0) ? lo : -32000.0;
            return (uint32_t)(int)cl;
        };
        auto flush_tile = [&](uint32_t tileStart, uint32_t validTo) __attribute__((always_inline)) {
            rowCount[lane] = f.produced;
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // this wave alone owns the tile: wave-level ordering is enough
            constexpr int kChunksPerRow = kTile / 8;
            constexpr int kRowsPerPass = kLanes / kChunksPerRow;
#pragma unroll
            for (int p = 0; p < kLanes / kRowsPerPass; ++p) {
                const int row = p * kRowsPerPass + lane / kChunksPerRow;
                const int chunk = lane % kChunksPerRow;
                const uint2* src = reinterpret_cast<const uint2*>(tile + row * kTileStride + chunk * 16);
                const uint2 lo = src[0], hi = src[1];
                const uint32_t first = tileStart + (uint32_t)chunk * 8u;
                if (rowCount[row] > first && first < validTo) {
                    uint4* dst = reinterpret_cast<uint4*>(A.pcm + rowBase[row] + first);
                    *dst = make_uint4(lo.x, lo.y, hi.x, hi.y);
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        };

        uint32_t it = 0;
        for (int iter = 0; iter < nIter; ++iter) {
            STAMP_BEGIN();
            const int c = iter - lag;
            if (c >= 0 && c < nChunks) {
                const uint32_t tpos = it % kTile;
                const int kind = chunk_kind(f);
                STAMP_KIND(kind);
                if (kind == 0) {
                    if (!f.done) {
_Pragma(KLATT_STR(unroll KLATT_UNROLL))
                        for (int i = 0; i < kChunk; ++i) {
                            const double o = NOISE ? PIPE(pipeO, c, i) : PIPE(pipeA, c, i);
                            const double y = NOISE ? PIPE(pipeA, c, i) : 0.0;
                            const double pt = NOISE ? PIPE(pipeB, c, i) : 0.0;
                            myRow[tpos + i] = (int16_t)finish(o, y, pt);
                        }
                        f.cnt += kChunk; f.produced += kChunk;
                    }
                } else if (kind == 1) {
                    const uint32_t wPar = wave_or(f.parMask), wRes = wave_or(f.resMask);
                    if (!f.done) {
#pragma nounroll
                        for (int i = 0; i < kChunk; ++i) {
                            stage_fade_masked<D, MODE, false>(f, nullptr, A, RF, RB, wPar, wRes);
                            const double o = NOISE ? PIPE(pipeO, c, i) : PIPE(pipeA, c, i);
                            const double y = NOISE ? PIPE(pipeA, c, i) : 0.0;
                            const double pt = NOISE ? PIPE(pipeB, c, i) : 0.0;
                            myRow[tpos + i] = (int16_t)finish(o, y, pt);
                        }
                        f.produced += kChunk;
                    }
                } else {
#pragma nounroll
                    for (int i = 0; i < kChunk; ++i) {
                        const bool emit = stage_advance<D, MODE, false>(f, nullptr, nullptr, A, P, RF, RB, d, myFrames, myMeta);
                        if (emit) {
                            const double o = NOISE ? PIPE(pipeO, c, i) : PIPE(pipeA, c, i);
                            const double y = NOISE ? PIPE(pipeA, c, i) : 0.0;
                            const double pt = NOISE ? PIPE(pipeB, c, i) : 0.0;
                            myRow[tpos + i] = (int16_t)finish(o, y, pt);
                            f.produced++;
                        }
                    }
                }
                it += kChunk;
                if ((it % kTile) == 0) flush_tile(it - kTile, it);
            }
            STAMP_WORKED();
            __syncthreads();
            STAMP_SYNCED();
        }
        if ((it % kTile) != 0) flush_tile(it - (it % kTile), it);
    }
#ifdef KLATT_STAMPS
    if (A.debug && lane == 0) { unsigned long long* o = A.debug + (blockIdx.x * 4 + stage) * 8; o[0] = stWork; o[1] = stWait; o[2] = stFastN; o[3] = stFadeN; o[4] = stGenN; o[5] = stFastC; o[6] = stFadeC; o[7] = stGenC; }
#endif
#undef PIPE
}

}  // namespace klatt
