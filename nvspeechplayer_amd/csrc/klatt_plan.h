// klatt_plan.h -- what speechPlayer_batch_setUtterances needs to know about a FRAME before it can plan a batch, as one function of
// the frame's 47 values that runs on either side of the link.
//
// Planning a batch (klatt_engine.hip: classification of the utterances, plan_tracks) looked at every frame's 376 bytes twice on
// the host: once to classify (does the utterance need its noise sources, may it skip the nasal pair, are its parameters finite and
// within the range of klatt_math.h) and once to hash the 45 values a track depends on -- 0.6 GB per pass for BASELINE configs[2],
// most of the call's CPU time.  frame_facts() is both in ONE pass: a word of flags and a 128-bit hash of the shape values.  The
// host planner then works on 24 bytes per frame; when the frames arrive in page-locked memory (speechPlayer_hostAlloc) they cross
// the link first and klatt_frame_facts evaluates the same function where they land (HBM-bound: the frames are read once at the
// device's rate), so that the host never reads them at all.
// The hash stands for the values while a batch is planned: two frames with the same 128 bits are taken to hold the same 45 values
// (2^-128 per pair of honest frames).  It is not trusted blindly: every frame the planner recognised by its hash is compared, value
// for value, with the first frame that carried that hash -- on the device, where the frames are (klatt_verify_shared: one more read
// of the frames at HBM rate); a batch in which that comparison fails is planned again without tracks (klatt_engine.hip).
// Frames that arrive as RECORDS (speechPlayer_batch_setRecords) need neither: a record names its shape by number, frames with one
// number hold the same values by construction, and klatt_expand_frames builds the 376-byte frames in HBM from 32 bytes each.
#pragma once

#include <stdint.h>
#include <string.h>

#include "klatt_device.h"

namespace klatt {

struct FrameFacts {          // 24 B per frame
    unsigned long long h0, h1;   // hash of the 45 shape values (klatt_device.h, shape_param), bit patterns
    uint32_t flags;              // FACT_*
    uint32_t pad;
};
constexpr uint32_t FACT_NOISE = 1u;        // a noise gain is non-zero, or the parallel bank's coefficients may not be finite
constexpr uint32_t FACT_NONFINITE = 2u;    // some parameter is NaN or infinite
constexpr uint32_t FACT_NASAL = 4u;        // the nasal pair is coupled in, or could not be skipped safely
constexpr uint32_t FACT_UNBOUNDED = 8u;    // a frequency or bandwidth outside the range of klatt_math.h (the direct stages)

__host__ __device__ inline bool fact_finite(double v)
{
    unsigned long long w;
    memcpy(&w, &v, 8);
    return (w & 0x7FF0000000000000ull) != 0x7FF0000000000000ull;
}
__host__ __device__ inline double fact_abs(double v) { return v < 0 ? -v : v; }      // (NaN stays NaN: every test below is written to fail on it)
// key n of the hash (splitmix64 of n: a constant wherever n is one)
__host__ __device__ constexpr unsigned long long fact_key(int n)
{
    unsigned long long z = (unsigned long long)(n + 1) * 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
// high ^ low half of the 128-bit product, plus both operands: a zero operand (a value whose bit pattern equals its key) annihilates
// the product, not the word -- the other operand still counts
__host__ __device__ inline unsigned long long fact_fold(unsigned long long a, unsigned long long c)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return (__umul64hi(a, c) ^ (a * c)) + a + ((c << 32) | (c >> 32));
#else
    const unsigned __int128 m = (unsigned __int128)a * c;
    return ((unsigned long long)(m >> 64) ^ (unsigned long long)m) + a + ((c << 32) | (c >> 32));
#endif
}

// what the two pitches (parameters 0 and 46) contribute to a frame's flags
__host__ __device__ inline uint32_t pitch_flags(double p0, double p46)
{
    uint32_t fl = 0;
    if (!fact_finite(p0) || !fact_finite(p46)) fl |= 2u;                         // FACT_NONFINITE
    if (!(fact_abs(p0) <= 1e30) || !(fact_abs(p46) <= 1e30)) fl |= 4u;           // FACT_NASAL: the source must stay bounded
    return fl;
}
// ... and what parameters 1..45 contribute (p: all 47; 0 and 46 are not looked at)
__host__ __device__ inline uint32_t shape_flags(const double* p, double maxF, double maxBw)
{
    uint32_t fl = 0;
    // Noise sources and the parallel bank can be skipped for an utterance only if every frame has all three noise gains exactly zero
    // (voiceTurbulenceAmplitude, aspirationAmplitude, fricationAmplitude) and no non-finite parameter that could turn 0 * x into NaN.
    if (p[3] != 0.0 || p[6] != 0.0 || p[24] != 0.0) fl |= 1u;                    // FACT_NOISE
    bool finite = true;
    for (int i = 1; i < kNumParams - 1; ++i) finite = finite && fact_finite(p[i]);
    if (!finite) fl |= 2u;
    // The skipped parallel bank contributes exactly 0 only while its coefficients are finite (a * 0 with a = inf is NaN, which the
    // reference clips to 32000): bandwidths in [0, 1e6], bounded frequencies (reference src/speechWaveGenerator.cpp:112-127).
    for (int i = 25; i <= 30; ++i)
        if (!(fact_abs(p[i]) <= 1e6) || !(p[i + 6] >= 0.0) || !(p[i + 6] <= 1e6)) fl |= 1u;
    // The nasal pair N0 -> NP enters the cascade as lerp(x, np, caNP) (reference src/speechWaveGenerator.cpp:151-152): with caNP == 0 in
    // every frame that is x as long as np stays finite -- bounded source, N0's zero pair not degenerate, NP not growing.
    if (p[23] != 0.0 || !(p[21] >= 1.0) || !(p[22] >= 0.0) || !(p[21] <= 1e6) || !(p[22] <= 1e6) ||
        !(fact_abs(p[13]) <= 1e6) || !(fact_abs(p[14]) <= 1e6) || !(fact_abs(p[5]) <= 1e30) || !(fact_abs(p[44]) <= 1e30))
        fl |= 4u;
    // the direct stages evaluate exp / cos with klatt_math.h alone, whose range is |arg| <= 700 / 1e4: frequencies (parameters 7..14,
    // 25..30) and bandwidths (15..22, 31..36) bounded accordingly
    bool inRange = true;
    for (int i = 7; i <= 14; ++i) inRange = inRange && (fact_abs(p[i]) <= maxF) && (fact_abs(p[i + 8]) <= maxBw);
    for (int i = 25; i <= 30; ++i) inRange = inRange && (fact_abs(p[i]) <= maxF) && (fact_abs(p[i + 6]) <= maxBw);
    if (!inRange) fl |= 8u;                                                      // FACT_UNBOUNDED
    return fl;
}
// are the 45 shape values (parameters 1..45) of two frames the same bits?
__host__ __device__ inline bool shape_values_equal(const double* a, const double* c)
{
    bool same = true;
    for (int i = 1; i < kNumParams - 1; ++i) {
        unsigned long long x, y;
        memcpy(&x, &a[i], 8); memcpy(&y, &c[i], 8);
        same = same && x == y;
    }
    return same;
}

// p: the frame's 47 parameters (include/speechPlayer.h); maxF / maxBw: the direct stages' bounds for this sample rate
__host__ __device__ inline FrameFacts frame_facts(const double* p, double maxF, double maxBw)
{
    FrameFacts o;
    const uint32_t fl = shape_flags(p, maxF, maxBw) | pitch_flags(p[0], p[46]);
    // The 45 shape values are parameters 1..45 (klatt_device.h, shape_param: every parameter but the two pitches).  Two independently
    // keyed multiply-fold sums over them, a pair of values per 64 x 64 -> 128-bit product (every position has keys of its own: the
    // same values in other places are another frame), each then avalanched: ~23 multiplies per word instead of a chain of 45.
    unsigned long long x = 0x9E3779B97F4A7C15ull, y = 0xC2B2AE3D27D4EB4Full;
#pragma unroll
    for (int i = 0; i < 46; i += 2) {      // (unrolled: the keys are literals)
        unsigned long long w0, w1 = 0;
        memcpy(&w0, &p[1 + i], 8);
        if (i + 1 < kShapeValues) memcpy(&w1, &p[2 + i], 8);
        x ^= fact_fold(w0 ^ fact_key(2 * i), w1 ^ fact_key(2 * i + 1));
        y ^= fact_fold(w0 ^ fact_key(2 * i + 100), w1 ^ fact_key(2 * i + 101));
    }
    x ^= x >> 32; x *= 0xD6E8FEB86659FD93ull; x ^= x >> 32;
    y ^= y >> 31; y *= 0xBF58476D1CE4E5B9ull; y ^= y >> 29;
    o.h0 = x; o.h1 = y; o.flags = fl; o.pad = 0;
    return o;
}

#if defined(__HIPCC__)
// one thread per frame: 376 bytes in, 24 out (HBM-bound; a wavefront's loads cover 24 KB of consecutive frames)
__global__ void __launch_bounds__(256) klatt_frame_facts(const double* __restrict__ frames, FrameFacts* __restrict__ out, long long nFrames, double maxF, double maxBw)
{
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long k = (long long)blockIdx.x * blockDim.x + threadIdx.x; k < nFrames; k += stride) {
        double p[kNumParams];
        const double* src = frames + k * kNumParams;
#pragma unroll
        for (int i = 0; i < kNumParams; ++i) p[i] = src[i];
        out[k] = frame_facts(p, maxF, maxBw);
    }
}

// A frame as a producer knows it (include/speechPlayer_batch.h, speechPlayer_frameRecord_t): 32 bytes
struct FrameRecord {
    double voicePitch, endVoicePitch;
    uint32_t shape;              // row of the shape table; kRecordSilence: the call passed framePtr == NULL
    uint32_t minSamples, fadeSamples;
    int32_t userIndex;
};
constexpr uint32_t kRecordSilence = 0xFFFFFFFFu;
static_assert(sizeof(FrameRecord) == 32, "FrameRecord layout");

// records -> the frames and their meta words in HBM: frame k = row records[k].shape of the shape table with parameters 0 and 46 from
// the record; silence = zeros, flagged.  A workgroup takes TILES of 64 frames: their records go through LDS once (one 32-byte record per
// thread, which also writes the frame's 16-byte meta word), then the tile's 3008 doubles leave as 16-byte stores, consecutive threads on
// consecutive addresses (a tile starts on a 16-byte boundary: 64 x 376 bytes); the shape table is a few hundred rows and stays in L2.
// HBM-bound: 32 B read, 392 B written per frame (1.58 M frames: 0.30 ms as one thread per double with a 64-bit division each, round 6's
// first form; this form: see profiles/r6_set_kernels.txt).
__global__ void __launch_bounds__(256) klatt_expand_frames(const FrameRecord* __restrict__ records, const double* __restrict__ shapes,
                                                           double* __restrict__ frames, FrameMeta* __restrict__ meta, long long nFrames)
{
    constexpr int kTileFrames = 64, kTilePairs = kTileFrames * kNumParams / 2;
    __shared__ FrameRecord rec[kTileFrames];
    const long long nTiles = (nFrames + kTileFrames - 1) / kTileFrames;
    for (long long tile = blockIdx.x; tile < nTiles; tile += gridDim.x) {
        const long long k0 = tile * kTileFrames;
        const int n = (int)((nFrames - k0) < (long long)kTileFrames ? (nFrames - k0) : (long long)kTileFrames);
        __syncthreads();      // (the previous tile's readers are done with `rec`)
        if ((int)threadIdx.x < n) {
            const FrameRecord r = records[k0 + threadIdx.x];
            rec[threadIdx.x] = r;
            FrameMeta m;
            m.minSamples = r.minSamples;
            m.fadeSamples = r.fadeSamples > 1u ? r.fadeSamples : 1u;      // reference src/speechPlayer.cpp:36
            m.userIndex = r.userIndex;
            m.flags = r.shape == kRecordSilence ? FRAME_NULL : 0u;
            meta[k0 + threadIdx.x] = m;
        }
        __syncthreads();
        auto value = [&](int j) -> double {      // double j of the tile
            const int f = j / kNumParams, p = j - f * kNumParams;
            if (f >= n) return 0.0;
            const uint32_t shape = rec[f].shape;
            if (shape == kRecordSilence) return 0.0;
            return p == 0 ? rec[f].voicePitch : (p == kNumParams - 1 ? rec[f].endVoicePitch : shapes[(long long)shape * kNumParams + p]);
        };
        double2* const out = reinterpret_cast<double2*>(frames + k0 * kNumParams);
        const int pairs = (n * kNumParams + 1) / 2;
        for (int j = (int)threadIdx.x; j < kTilePairs && j < pairs; j += (int)blockDim.x) {
            const double x = value(2 * j), y = value(2 * j + 1);
            if (2 * j + 1 < n * kNumParams) out[j] = make_double2(x, y);
            else frames[k0 * kNumParams + 2 * j] = x;      // (the odd last double of a short last tile)
        }
    }
}

// rep[k]: the frame whose 45 shape values the planner took frame k's to be (the first frame it saw with k's hash), or k itself /
// 0xFFFFFFFF for a frame nobody stood in for.  One thread per frame; any difference sets *mismatch (and the first frame found).
__global__ void __launch_bounds__(256) klatt_verify_shared(const double* __restrict__ frames, const uint32_t* __restrict__ rep, long long nFrames,
                                                           unsigned long long* __restrict__ mismatch)
{
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long k = (long long)blockIdx.x * blockDim.x + threadIdx.x; k < nFrames; k += stride) {
        const uint32_t r = rep[k];
        if (r == 0xFFFFFFFFu || (long long)r == k) continue;
        if (!shape_values_equal(frames + k * kNumParams, frames + (long long)r * kNumParams))
            atomicMin(mismatch, (unsigned long long)k);
    }
}
#endif

}  // namespace klatt
