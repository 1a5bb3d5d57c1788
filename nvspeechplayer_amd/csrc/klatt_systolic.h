// klatt_systolic.h -- stage-parallel batch kernel: one workgroup = 4 wavefronts = 4 pipeline stages
// over the SAME 64 utterances (lane = utterance in every wave).
//
// The lane-per-utterance kernel (klatt_device.h) is bound by f64 VALU issue of ONE wave per 64
// utterances and by its register/LDS footprint (3 waves per CU).  The per-sample work of the
// reference's generate() loop (reference src/speechWaveGenerator.cpp:197-214) is a chain
//     source -> N0 -> NP -> r6..r1 -> (+ parallel bank) -> gain/clip
// whose links only pass one double per sample.  Here the links are cut into four stages that run
// on the four SIMDs of a CU at the same time, each on a different chunk of kChunk samples, handing
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
// Every stage runs its own copy of the frame state machine (reference src/frame.cpp:41-80) for just
// the parameters it needs: the fade's old/new values in its own LDS region ([param][lane]), the
// current values, coefficients and filter memories in registers.  No operation, operand or
// rounding differs from the lane kernel / the reference: the stages compute the same values in the
// same order, only on different SIMDs.  A 4096-utterance batch becomes 256 wavefronts instead of 64.
//
// Barrier discipline: the four waves take four different branches (stage = wave index), and every branch
// runs the same loop `for iter < nIter` with exactly one __syncthreads() per iteration; nIter comes from the
// longest utterance of the group (LDS atomicMax before the loops), so every wave reaches every barrier
// whether or not it has a chunk to process.  A stage of depth d works on chunk iter - d: it reads what its
// producer wrote in the previous iteration into buffer (chunk & 1) while the producer fills the other one.
// The PCM tile belongs to the final stage's wave alone (wave-level ordering, no barrier).
#pragma once

#include <type_traits>

#include "klatt_device.h"

namespace klatt {

#ifndef KLATT_UNROLL
#define KLATT_UNROLL 8
#endif
#ifndef KLATT_MINWAVES
#define KLATT_MINWAVES 1
#endif
#ifndef KLATT_PAIR
#define KLATT_PAIR 1
#endif
#ifndef KLATT_NOISY_RUNS
#define KLATT_NOISY_RUNS 0      // 1: uniform runs inside event chunks for the noisy kernels too (measured slower: cfg2 15.8 -> 16.8 ms)
#endif
#ifndef KLATT_FADE_TIGHT
#define KLATT_FADE_TIGHT 1      // a fade's chunks run in a tight loop (what moves is fixed for the fade): cfg2 16.4 -> 15.7 ms; 0 decides chunk by chunk
#endif
#ifndef KLATT_NOISY_TIGHT
#define KLATT_NOISY_TIGHT 1     // the noisy kernels run steady stretches in a tight loop too (cfg2 20.4 -> 16.2 ms); 0 decides chunk by chunk
#endif
#define KLATT_STR2(x) #x
#define KLATT_STR(x) KLATT_STR2(x)

#ifndef KLATT_FLAT_EXHAUSTIVE
#define KLATT_FLAT_EXHAUSTIVE 1 // flat launches: the final stage is the chain's unconditional last branch, so that the compiler sees that a flat launch
                                // runs none of the untracked stages (it cannot tell that a stage number is 0..3): the kernel is half the code, 245
                                // VGPRs instead of 256 and NO scratch instead of 128 bytes per lane (0: the test `stage == 2`, as before)
#endif
constexpr int kStages = 4;
#ifndef KLATT_FLAT_SOURCE
#define KLATT_FLAT_SOURCE 1     // flat launches: S0 is a flat stage too (0: the source stage of the noisy launches, with its frame state machine)
#endif

// LDS per workgroup: pipes [2 buffers][CH][64 lanes] f64 (4 noisy / 3 quiet), the PCM tile and row
// info of the final stage, then each stage's old/new parameter region.  CH = samples per pipeline hand-over.
// SPLIT: which stage runs what (stream_split below): bits 0-1 cascade resonators (r3, then r2, then r1) that the cascade's head stage takes
// over from the final stage, bit 2 the nasal pair N0, NP in the source stage instead of the head stage
template <bool NOISE, int CH, bool FLAT = false, int SPLIT = 0>
struct SysLds {
    static constexpr int MOVED = SPLIT & 3;
    static constexpr bool NP0 = (SPLIT >> 2) != 0;
    static constexpr int kBufs = 2;                                        // buffers of a pipe (the barrier keeps the stages within one chunk of each other)
    static constexpr int kBufsX = 2;                                       // ... of pipe X (S0 -> S1), the first in memory
    static constexpr int kBufBytes = CH * kLanes * (FLAT ? (int)sizeof(sig_t) : 8);
    static constexpr int kPipeBytes = kBufs * kBufBytes;
    static constexpr int kNumPipes = NOISE ? 4 : 3;
    static constexpr int kPipeO = kBufsX * kBufBytes;                      // the pipes in memory: X, O, A(, B)
    static constexpr int kTileOff = kPipeO + (kNumPipes - 1) * kPipeBytes;
    static constexpr int kRowBase = kTileOff + kLanes * kTileStride;
    static constexpr int kRowCount = kRowBase + kLanes * 8;
    static constexpr int kMaxLen = kRowCount + kLanes * 4;
    static constexpr int kSync = kMaxLen + 16;         // eight words the waves use while they pick their stages
    static constexpr int kFrames = kSync + 32;
    // parameters per stage (S0, S1, S2, S3): noisy 7, 11, 14, 14; quiet 7, 7, 6, 5
    // flat launches (FLAT): the stages take everything from the tracks and keep no fade end points: 0, 0, 0, 0
    static constexpr int kParams0 = (FLAT && KLATT_FLAT_SOURCE) ? 0 : (NP0 ? 12 : 7), kParams1 = FLAT ? 0 : (NOISE ? (NP0 ? 0 : 5) + 2 * (3 + MOVED) : 7), kParams2 = FLAT ? 0 : (NOISE ? 14 - 2 * MOVED : 6), kParams3 = FLAT ? 0 : (NOISE ? 14 : 5);
    static constexpr int kFrames1 = kFrames + 1 * kParams0 * kLanes * 8;    // S0 keeps its target values in registers
    static constexpr int kFrames2 = kFrames1 + 2 * kParams1 * kLanes * 8;
    static constexpr int kFrames3 = kFrames2 + 2 * kParams2 * kLanes * 8;
    static constexpr int kBytes = kFrames3 + 2 * kParams3 * kLanes * 8;
};

// ---- the frame state machine, restricted to a stage's parameter subset ----------------------
// NPARAM tracked parameters P[0..NPARAM); resonator r uses (P[RF[r]], P[RB[r]]).
template <int NPARAM, int NRES, bool NWREG = false>
struct StageFrame {
    double* oldL;     // LDS [NPARAM][64]: this lane's slot k at oldL[k * 64]
    double* nwL;      // the fade's target values: LDS like oldL, or (NWREG, small stages) registers
    double nwR[NWREG && NPARAM > 0 ? NPARAM : 1];
    __device__ __forceinline__ double getNew(int k) const { return NWREG ? nwR[k] : nwL[k * kLanes]; }
    __device__ __forceinline__ void setNew(int k, double v) { if (NWREG) nwR[k] = v; else nwL[k * kLanes] = v; }
    double cur[NPARAM > 0 ? NPARAM : 1];
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

template <int NPARAM, int NRES, bool NWREG>
__device__ __forceinline__ void stage_frame_init(StageFrame<NPARAM, NRES, NWREG>& f, bool live, unsigned char* region, int lane)
{
    f.oldL = reinterpret_cast<double*>(region) + lane;
    f.nwL = f.oldL + NPARAM * kLanes;
#pragma unroll
    for (int k = 0; k < NPARAM; ++k) { f.oldL[k * kLanes] = 0.0; f.setNew(k, 0.0); f.cur[k] = 0.0; }
#pragma unroll
    for (int r = 0; r < NRES; ++r) { f.ra[r] = 0.0; f.rb[r] = 2.0; f.rc[r] = -1.0; f.z1[r] = 0.0; f.z2[r] = 0.0; }
    f.invFade = 1.0;
    f.cnt = 0; f.oldMin = 0; f.newMin = 0; f.newFade = 1; f.nextFrame = 0; f.resMask = 0; f.parMask = 0; f.produced = 0;
    f.hasNew = false; f.oldNull = true; f.newNull = false; f.done = !live;
}

// Stage descriptor.  GAIN = index into P of parameter 44 (or -1): NULL frames force it to 0
// (reference src/frame.cpp:61,66).  ANTI0: resonator 0 is the anti-resonator N0.
template <int NPARAM_, int NRES_, int GAIN_, bool PITCH_, bool ANTI0_, bool INLINE_COEF_ = false>
struct StageDesc {
    static constexpr int NPARAM = NPARAM_, NRES = NRES_, GAIN = GAIN_;
    static constexpr bool PITCH = PITCH_, ANTI0 = ANTI0_, INLINE_COEF = INLINE_COEF_;
};

struct StageCtx {          // what every stage needs from the launch
    const KernelArgs& A;
    const UttDesc& d;
    const double* myFrames;
    const FrameMeta* myMeta;
    uint32_t ringOff, ringMask;    // frame k of the queue is myFrames[(ringOff + k) & ringMask] (frame_window, klatt_device.h)
    const FlatRef* myFlat;     // flat launches: the utterance's per-frame track references, loaded ahead by the stages (klatt_device.h)
    const SourceRef* mySrc;    // flat launches: what the source stage loads ahead
    bool lone = false;         // live handles: every lane of the wavefront advances ONE handle (streams_synthesize's replicas); wave-uniform
    double* loneLds = nullptr;   // ... and this stage's 12 KB of LDS behind the kernel's own (kLoneLdsPerStage): the values of a fade's next 64 samples, [slot][sample]
};
constexpr int kLoneLdsPerStage = 12288;
// Live handles with one workgroup per CU (up to 16 384 of them; one alone in its wavefront): a launch's pace is its slowest stage's, not a
// SIMD's load, and the batch launches' split -- source | N0, NP, r6..r4 | r3..r1, parallel 5, 6, mix, PCM | frication, parallel 1..4: a lone
// handle's steady chunk 2570 | 3360 | 4960 | 3970 ticks -- is balanced for pairs of stages on a SIMD.  SPLIT (SysLds above): of the final
// stage's three cascade resonators bits 0-1 go to the cascade's head stage; bit 2: the nasal pair N0, NP goes from the head stage to the
// source stage.  7: source, N0, NP | r6..r1 | parallel 5, 6, mix, PCM | frication, parallel 1..4 = 3530 | 3310 | 3930 | 3850 (one pull of a
// lone handle 1.46 -> 1.37 (split 1) -> 1.33 (2) -> 1.30 ms (7); 8192 handles in step 2.53 -> 2.48 ms; unrelated 13.3 -> 12.3).  Same
// arithmetic in the same order; a handle's saved state is numbered by resonator, not by stage, so pulls may alternate between splits.
// Unrelated handles that share wavefronts run every chunk sample by sample, where a stage's cost follows the resonators it evaluates:
// split 5 -- source, N0, NP | r6..r3 | r2, r1, parallel 5, 6, mix, PCM | frication, parallel 1..4 -- 11.2 ms per pull of 8192 handles
// against 11.6 with split 7, 11.3 with 6, 12.7 with 4.
#ifndef KLATT_STREAM_SPLIT
#define KLATT_STREAM_SPLIT 5      // live handles sharing wavefronts (one workgroup per CU), see below
#endif
#ifndef KLATT_LONE_SPLIT
#define KLATT_LONE_SPLIT 7        // a handle alone in its wavefront
#endif
// the noisy final stage's parameters and resonators (block numbering) once `moved` cascade resonators have left it
__device__ constexpr int final_param(int moved, int k) { constexpr int L0[14] = {9, 17, 8, 16, 7, 15, 29, 35, 30, 36, 41, 42, 43, 45}; return L0[k + 2 * moved < 13 ? k + 2 * moved : 13]; }
__device__ constexpr int final_res(int moved, int r) { constexpr int G0[5] = {5, 6, 7, 12, 13}; return G0[r + moved < 4 ? r + moved : 4]; }
template <bool STREAM, int WPS, bool LONE> constexpr int stream_split() { return (STREAM && WPS == 1) ? (LONE ? KLATT_LONE_SPLIT : KLATT_STREAM_SPLIT) : 0; }
// the parameters of the cascade's head stage: (f, bw) of [N0, NP,] r6 .. (ncasc resonators), [then caNP]
__device__ constexpr int head_param(bool nasal, int ncasc, int k)
{
    constexpr int N[4] = {13, 21, 14, 22}, C[12] = {12, 20, 11, 19, 10, 18, 9, 17, 8, 16, 7, 15};
    if (nasal && k < 4) return N[k];
    const int j = nasal ? k - 4 : k;
    return j < 2 * ncasc ? C[j] : 23;
}

// OR over the wavefront of the low NBITS bits of a per-lane mask, as a wave-uniform (scalar) value.
// One ballot per bit: a handful of instructions, no LDS round trips (a shuffle reduction costs 6).
template <int NBITS>
__device__ __forceinline__ uint32_t wave_or_bits(uint32_t m)
{
    uint32_t w = 0;
#pragma unroll
    for (int k = 0; k < NBITS; ++k) w |= __any(m & (1u << k)) ? (1u << k) : 0u;
    return w;
}

// parMask bit 31: some target of the running fade is NaN ("hold", reference src/utils.h:21).  When no live lane of
// the wave has the bit, the whole-chunk fade paths interpolate without the NaN test (three instructions per parameter and sample);
// with it the chunk goes sample by sample.
constexpr uint32_t kNanTarget = 0x80000000u;
// wave-uniform: some live lane fades toward a NaN target
template <class SF>
__device__ __forceinline__ bool nan_target_live(const SF& f) { return __any(!f.done && (f.parMask & kNanTarget) != 0u); }

// one event sample (fade end / dequeue / end of queue) for a stage; mirrors event_step() of klatt_device.h
template <class D, class SF>
__device__ __forceinline__ bool stage_event(SF& f, PitchState* ps, int32_t* lastIndex, const int* P, const int* RF, const int* RB,
                                            const StageCtx& X)
{
    if (f.hasNew) {   // fade finished (reference src/frame.cpp:44-47)
#pragma unroll
        for (int k = 0; k < D::NPARAM; ++k) f.oldL[k * kLanes] = f.getNew(k);
        f.oldMin = f.newMin; f.oldNull = f.newNull;
        if (D::PITCH) { ps->old0 = ps->new0; ps->oldInc = ps->newInc; }
        f.hasNew = false;
        return true;
    }
    if (f.nextFrame >= X.d.nFrames) { f.done = true; return false; }   // queue empty (:74)
    const uint32_t at = (X.ringOff + f.nextFrame) & X.ringMask;
    const FrameMeta m = X.myMeta[at];
    const double* g = X.myFrames + (size_t)at * kNumParams;
    f.nextFrame++;
    f.newMin = m.minSamples; f.newFade = m.fadeSamples; f.newNull = (m.flags & FRAME_NULL) != 0;
    constexpr int GI = D::GAIN >= 0 ? D::GAIN : 0;
    double pNew0 = 0.0, pNewInc = 0.0, pOld0 = D::PITCH ? ps->old0 : 0.0;
    if (f.newNull) {   // silence keeps the old shape, gain gated off (:59-63)
#pragma unroll
        for (int k = 0; k < D::NPARAM; ++k) f.setNew(k, f.oldL[k * kLanes]);
        uint32_t pm = 0;
        if (D::GAIN >= 0) { pm = (f.oldL[GI * kLanes] != 0.0) ? (1u << GI) : 0u; f.setNew(GI, 0.0); }
        if (D::PITCH) { pNew0 = ps->cur0; pNewInc = 0.0; }
        f.resMask = 0; f.parMask = pm;
    } else {
        uint32_t pm = 0, mk = 0;
        if (f.oldNull) {   // coming out of silence: start from the new shape, gain 0 (:64-67)
#pragma unroll
            for (int k = 0; k < D::NPARAM; ++k) { const double v = g[P[k]]; f.setNew(k, v); f.oldL[k * kLanes] = v; pm |= (v != v) ? kNanTarget : 0u; }
            // nothing but the gain moves (old == new elsewhere); a NaN target ("hold", src/utils.h:21) keeps its flag so that
            // the whole-chunk fade paths, which interpolate without the NaN test, stay away from this fade
            if (D::GAIN >= 0) { pm = (pm & kNanTarget) | ((f.getNew(GI) != 0.0) ? (1u << GI) : 0u); f.oldL[GI * kLanes] = 0.0; }
        } else {
            bool moved[D::NPARAM > 0 ? D::NPARAM : 1];
#pragma unroll
            for (int k = 0; k < D::NPARAM; ++k) {
                const double v = g[P[k]];
                moved[k] = !(v == f.oldL[k * kLanes]);      // NaN ("hold") counts as moving: harmless
                f.setNew(k, v);
                pm |= moved[k] ? (1u << k) : 0u;
                pm |= (v != v) ? kNanTarget : 0u;
            }
#pragma unroll
            for (int r = 0; r < D::NRES; ++r) mk |= (moved[RF[r]] || moved[RB[r]]) ? (1u << r) : 0u;
        }
        if (D::PITCH) {
            const double g0 = g[0], g46 = g[46];
            pNew0 = g0;
            pNewInc = (g46 - g0) / (double)f.newMin;   // reference src/frame.cpp:98
            if (f.oldNull) pOld0 = g0;
        }
        f.resMask = mk; f.parMask = pm;
    }
    // (one store per field after the branches: stores to different fields in different branches are merged by the optimiser into
    // one store through a selected POINTER, which keeps the whole PitchState in scratch memory)
    if (D::PITCH) { ps->new0 = pNew0; ps->newInc = pNewInc; ps->old0 = pOld0; }
    if (lastIndex && m.userIndex != -1) *lastIndex = m.userIndex;   // (:69)
    f.cnt = 0;                                                       // (:70)
    if (D::PITCH) {
        ps->new0 += ps->newInc * (double)f.newFade;                  // (:71)
        f.parMask |= (ps->new0 != ps->new0) ? kNanTarget : 0u;
    }
    f.invFade = 1.0 / (double)f.newFade;
    f.hasNew = true;
    return true;
}

// One fade sample's update (reference src/frame.cpp:48-53).  `lerp` and `wRes` are WAVE-UNIFORM:
//   lerp  : interpolate this stage's parameters (all of them: their LDS loads go out back to back and
//           cost one wait, which beats skipping individual parameters behind serialised waits);
//           false only when NO lane's parameters move, then every `cur` already holds its value;
//   wRes  : resonators whose (f, bw) move in SOME lane.
// For a lane whose own old == new the update recomputes the value it already holds (old + 0*ratio;
// coefficients are a pure function of (f, bw), reference :112-127), so wave-level decisions give
// every lane exactly what the reference computes.
// Classes of the stage's resonators for one whole fade, two bits each (COEF_*), from the fade's end points: the
// interpolated (f, bw) stay between them, so if both ends of every live lane need no reduction (or sit in quadrant -1),
// with a margin against the one-ulp overshoot of `from + (to - from) * ratio`, every sample of the fade does.  Wave-uniform.
constexpr uint32_t kCoefAllUnknown = 0xAAAAAAAAu;
template <class D, class SF>
__device__ __forceinline__ uint32_t fade_classes(const SF& f, const KernelArgs& A, const int* RF, const int* RB, uint32_t wRes)
{
    uint32_t bits = kCoefAllUnknown;
    if (!D::INLINE_COEF) return bits;
#pragma unroll
    for (int r = 0; r < D::NRES; ++r) {
        if (!(wRes & (1u << r))) continue;
        const double xo = A.negPiOverSr * f.oldL[RB[r] * kLanes] * kLog2e, xn = A.negPiOverSr * f.getNew(RB[r]) * kLog2e;
        const double to = A.twoPiOverSr * -f.oldL[RF[r] * kLanes] * kTwoOverPi, tn = A.twoPiOverSr * -f.getNew(RF[r]) * kTwoOverPi;
        const bool eu = __builtin_fabs(xo) <= 0.499 && __builtin_fabs(xn) <= 0.499;
        const bool c0 = __builtin_fabs(to) <= 0.499 && __builtin_fabs(tn) <= 0.499;
        const bool c1 = to <= -0.501 && to >= -1.499 && tn <= -0.501 && tn >= -1.499;
        const uint32_t cls = __all(f.done || (eu && c0)) ? COEF_UNREDUCED : (__all(f.done || (eu && c1)) ? COEF_QUADRANT_M1 : COEF_UNKNOWN);
        bits = (bits & ~(3u << (2 * r))) | (cls << (2 * r));
    }
    return bits;
}

// value of lane `srcLane` (wave-uniform) in every lane
__device__ __forceinline__ double lane_read(double v, int srcLane)
{
    const unsigned long long w = (unsigned long long)__double_as_longlong(v);
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)w, srcLane), hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(w >> 32), srcLane);
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}

template <class D, int MODE, bool PLAIN = false, class SF>
__device__ __forceinline__ void stage_fade(SF& f, PitchState* ps, const KernelArgs& A, const int* RF, const int* RB,
                                           bool lerp, uint32_t wRes, bool gainOnly = false, uint32_t coefCls = kCoefAllUnknown,
                                           const double* co = nullptr, const double* cn = nullptr)
{
    if (!D::PITCH && !lerp) return;
    const double ratio = div_by((double)f.cnt, (double)f.newFade, f.invFade);
    if (D::PITCH) ps->cur0 = fade_value(ps->old0, ps->new0, ratio);
    constexpr int GI = D::GAIN >= 0 ? D::GAIN : 0;
    if (D::GAIN >= 0 && lerp && gainOnly) {
        // fades into and out of silence move nothing but the gain (reference src/frame.cpp:59-67); `gainOnly` is wave-uniform
        f.cur[GI] = fade_value(f.oldL[GI * kLanes], f.getNew(GI), ratio);
    } else if (lerp) {
        double o[D::NPARAM > 0 ? D::NPARAM : 1], n[D::NPARAM > 0 ? D::NPARAM : 1];
#pragma unroll
        for (int k = 0; k < D::NPARAM; ++k) { o[k] = co ? co[k] : f.oldL[k * kLanes]; n[k] = cn ? cn[k] : f.getNew(k); }
#pragma unroll
        for (int k = 0; k < D::NPARAM; ++k) f.cur[k] = PLAIN ? o[k] + ((n[k] - o[k]) * ratio) : fade_value(o[k], n[k], ratio);   // PLAIN: no NaN target in any live lane
    }
#pragma unroll
    for (int r = 0; r < D::NRES; ++r) {
        if (wRes & (1u << r)) {
            // inlined where fades dominate (speech); the quiet kernels keep one out-of-line copy, which keeps
            // their loops small (measured: cfg1 1.72 ms vs 1.86 ms inlined; cfg2 29.2 ms inlined vs 36.6 ms called)
            const Coef k = D::INLINE_COEF
                ? resonator_coefficients_inline<MODE>(f.cur[RF[r]], f.cur[RB[r]], D::ANTI0 && r == 0, A.negPiOverSr, A.twoPiOverSr,
                                                      (int)((coefCls >> (2 * r)) & 3u))
                : resonator_coefficients<MODE>(f.cur[RF[r]], f.cur[RB[r]], D::ANTI0 && r == 0, A.negPiOverSr, A.twoPiOverSr);
            f.ra[r] = k.a; f.rb[r] = k.b; f.rc[r] = k.c;
        }
    }
}

// advance the state machine by one sample (any mix of lanes); returns true when a sample is emitted
template <class D, int MODE, class SF>
__device__ __forceinline__ bool stage_advance(SF& f, PitchState* ps, int32_t* lastIndex, const int* P, const int* RF, const int* RB,
                                              const StageCtx& X, const double* co = nullptr, const double* cn = nullptr, bool* anyEvent = nullptr)
{
    bool emit = false, ev = false;
    bool fading = false;
    if (!f.done) {
        f.cnt++;
        if (f.hasNew && f.cnt <= f.newFade) { fading = true; emit = true; }
        else if (!f.hasNew && f.cnt <= f.oldMin) {
            if (D::PITCH) { ps->cur0 += ps->oldInc; ps->old0 = ps->cur0; }   // glide (reference src/frame.cpp:76-79)
            emit = true;
        } else {
            emit = stage_event<D>(f, ps, lastIndex, P, RF, RB, X);
            ev = true;
        }
    }
    if (anyEvent) *anyEvent = __any(ev);
    if (__any(fading)) {
        // first fade sample of a lane: everything; later: what moves in some fading lane
        const bool lerp = __any(fading && (f.cnt == 1 || f.parMask != 0u));
        const uint32_t wRes = wave_or_bits<D::NRES>(fading ? ((f.cnt == 1) ? 0xFFFFFFFFu : f.resMask) : 0u);
        if (fading) stage_fade<D, MODE>(f, ps, X.A, RF, RB, lerp, wRes, false, kCoefAllUnknown, co, cn);
    }
    return emit;
}

// ---- live streams on the stage-parallel kernel ------------------------------------------------------------
// A handle's saved state is the 240-double block of klatt_device.h (the lane kernel's save block defines the layout).
// Every stage restores and saves ITS slice of it: its parameters' old / new / current values, its resonators'
// coefficients and memories, and its copy of the frame state machine's counters (one copy is saved: the stages' copies
// agree by construction).  Where parameter p's CURRENT value lives in the block:
__device__ constexpr int stream_cur_slot(int p)
{
    if (p == 0) return 137;
    for (int h = 0; h < kNumHot; ++h) if (kHot[h] == p) return 118 + h;
    for (int r = 0; r < kNumRes; ++r) { if (kResF[r] == p) return 90 + 2 * r; if (kResB[r] == p) return 91 + 2 * r; }
    return -1;
}
// P: the stage's parameters; GR: its resonators in the block's numbering (N0, NP, c6..c1, p1..p6 = 0..13)
template <class D, class SF>
__device__ __forceinline__ void stage_state_load(SF& f, PitchState* ps, const double* S, const int* P, const int* RF, const int* RB, const int* GR, bool purge)
{
    if (S[239] != 0.0) {
#pragma unroll
        for (int k = 0; k < D::NPARAM; ++k) {
            f.oldL[k * kLanes] = S[P[k] - 1];
            f.setNew(k, S[45 + P[k] - 1]);
            f.cur[k] = S[stream_cur_slot(P[k])];
        }
#pragma unroll
        for (int r = 0; r < D::NRES; ++r) {
            f.ra[r] = S[138 + GR[r]]; f.rb[r] = S[152 + GR[r]]; f.rc[r] = S[166 + GR[r]]; f.z1[r] = S[180 + GR[r]]; f.z2[r] = S[194 + GR[r]];
        }
        f.invFade = S[214];
        f.cnt = (uint32_t)S[215]; f.oldMin = (uint32_t)S[216]; f.newMin = (uint32_t)S[217]; f.newFade = (uint32_t)S[218];
        const uint32_t fl = (uint32_t)S[219];
        f.hasNew = fl & 1; f.oldNull = fl & 2; f.newNull = fl & 4;
        if (D::PITCH) { ps->old0 = S[135]; ps->new0 = S[136]; ps->cur0 = S[137]; ps->oldInc = S[212]; ps->newInc = S[213]; }
        // what moves in a running fade follows from its end points, as stage_event derived it
        uint32_t pm = 0, mk = 0;
        bool moved[D::NPARAM > 0 ? D::NPARAM : 1];
#pragma unroll
        for (int k = 0; k < D::NPARAM; ++k) {
            const double n = f.getNew(k);
            moved[k] = !(n == f.oldL[k * kLanes]);
            pm |= moved[k] ? (1u << k) : 0u;
            pm |= (n != n) ? kNanTarget : 0u;
        }
#pragma unroll
        for (int r = 0; r < D::NRES; ++r) mk |= (moved[RF[r]] || moved[RB[r]]) ? (1u << r) : 0u;
        if (D::PITCH) pm |= (ps->new0 != ps->new0) ? kNanTarget : 0u;
        f.parMask = f.hasNew ? pm : 0u;
        f.resMask = f.hasNew ? mk : 0u;
    }
    if (purge) {
        // reference src/frame.cpp:103-112: the queue was emptied on the host; cut over from the current interpolated frame
        f.cnt = f.oldMin;
        if (f.hasNew) {
            f.oldNull = f.newNull;
#pragma unroll
            for (int k = 0; k < D::NPARAM; ++k) f.oldL[k * kLanes] = f.cur[k];
            if (D::PITCH) ps->old0 = ps->cur0;
            f.hasNew = false;
            f.parMask = 0; f.resMask = 0;
        }
    }
}
template <class D, class SF>
__device__ __forceinline__ void stage_state_save(const SF& f, const PitchState* ps, double* S, const int* P, const int* GR)
{
#pragma unroll
    for (int k = 0; k < D::NPARAM; ++k) {
        S[P[k] - 1] = f.oldL[k * kLanes];
        S[45 + P[k] - 1] = f.getNew(k);
        S[stream_cur_slot(P[k])] = f.cur[k];
    }
#pragma unroll
    for (int r = 0; r < D::NRES; ++r) {
        S[138 + GR[r]] = f.ra[r]; S[152 + GR[r]] = f.rb[r]; S[166 + GR[r]] = f.rc[r]; S[180 + GR[r]] = f.z1[r]; S[194 + GR[r]] = f.z2[r];
    }
    if (D::PITCH) {
        S[135] = ps->old0; S[136] = ps->new0; S[137] = ps->cur0; S[212] = ps->oldInc; S[213] = ps->newInc;
        S[214] = f.invFade;
        S[215] = (double)f.cnt; S[216] = (double)f.oldMin; S[217] = (double)f.newMin; S[218] = (double)f.newFade;
        S[219] = (double)((f.hasNew ? 1u : 0u) | (f.oldNull ? 2u : 0u) | (f.newNull ? 4u : 0u));
        S[239] = 1.0;
    }
}

template <int MODE, class R>
__device__ __forceinline__ R resonate(R& z1, R& z2, R a, R b, R c, R in)
{
    const R y = dot3<MODE>(a, in, b, z1, c, z2);
    z2 = z1; z1 = y;
    return y;
}

// all live lanes steady (0) / all fading beyond their first fade sample (1), with at least n samples left; else -1
template <class SF>
__device__ __forceinline__ int run_kind(const SF& f, uint32_t n)
{
    const uint32_t rem = f.hasNew ? (f.newFade - f.cnt) : (f.oldMin > f.cnt ? f.oldMin - f.cnt : 0u);
    const bool roomy = f.done || rem >= n;
    if (!__all(roomy)) return -1;
    if (!__any(!f.done && f.hasNew)) return 0;
    if (!__any(!f.done && (!f.hasNew || f.cnt == 0))) return 1;
    return -1;
}
template <int CH, class SF>
__device__ __forceinline__ int chunk_kind(const SF& f) { return run_kind(f, (uint32_t)CH); }

// After chunk_kind() == 0: how many chunks (this one included) every live lane stays steady for, i.e. the minimum
// over the live lanes of (samples left in the steady stretch) / CH.  The chunks after the first need no new decision:
// nothing but the sample counter changes in a steady chunk (vibrato, which forces the sample-by-sample path, can only
// come alive at an event).  Wave-uniform result.
template <int CH, class SF>
__device__ __forceinline__ uint32_t steady_run(const SF& f)
{
    uint32_t n = f.done ? 0xFFFFFFFFu : (f.oldMin - f.cnt) / (uint32_t)CH;
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
        const uint32_t o = (uint32_t)__shfl_xor((int)n, m, kLanes);
        n = o < n ? o : n;
    }
    return (uint32_t)__builtin_amdgcn_readfirstlane((int)n);
}

// After chunk_kind() == 1: how many chunks (this one included) every live lane keeps fading for.
template <int CH, class SF>
__device__ __forceinline__ uint32_t fade_run(const SF& f)
{
    uint32_t n = f.done ? 0xFFFFFFFFu : (f.newFade - f.cnt) / (uint32_t)CH;
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
        const uint32_t o = (uint32_t)__shfl_xor((int)n, m, kLanes);
        n = o < n ? o : n;
    }
    return (uint32_t)__builtin_amdgcn_readfirstlane((int)n);
}

#ifdef KLATT_STAMPS
struct Stamps {
    unsigned long long work = 0, wait = 0, t0 = 0, t1 = 0, n[3] = {0, 0, 0}, c[3] = {0, 0, 0};
    int kind = 3;
};
// every chunk is counted under the kind its stretch was decided as (the chunks of a tight run keep the run's kind)
#define STAMP_BEGIN() st.t0 = __builtin_amdgcn_s_memtime()
#define STAMP_KIND(k) do { st.kind = (k) < 0 ? 2 : (k); } while (0)
#define STAMP_WORKED() do { st.t1 = __builtin_amdgcn_s_memtime(); st.work += st.t1 - st.t0; if (st.kind < 3) { st.c[st.kind] += st.t1 - st.t0; st.n[st.kind]++; } } while (0)
#define STAMP_IDLE() st.kind = 3
#define STAMP_SYNCED() do { st.wait += __builtin_amdgcn_s_memtime() - st.t1; } while (0)
#else
#define STAMP_BEGIN()
#define STAMP_IDLE()
#define STAMP_KIND(k)
#define STAMP_WORKED()
#define STAMP_SYNCED()
#endif

// ---- the chunk loop of one pipeline element (a wave) ---------------------------------------------------------------
// Barrier discipline: `nIter` iterations in every wave of the workgroup, one __syncthreads() each; a wave of depth d works on
// chunk iter - d.  Per chunk one of three wave-uniform paths: every live lane steady (straight-line, inputs optionally
// preloaded), every live lane fading, or sample by sample (events, mixed lanes, start-up skew, the partial last chunk of
// a launch of live handles).  Steady and fading stretches are decided once and then run in tight loops of their own (chunk,
// barrier, chunk, ...) with the barrier count of the outer loop.  The element's program is a set of callables:
//   forceGeneral()             true: this chunk goes sample by sample whatever the lanes do (vibrato live, start-up skew)
//   begin(kind)                wave-uniform set-up before a chunk's samples; true selects altChunk for a steady stretch
//   altChunk(c)                a whole steady chunk, the element's own straight-line version (elements without K::PRE)
//   altSample(c, i, pre)       one sample of such a chunk, input preloaded (elements with K::PRE)
//   preIn(c, i)                LDS input of sample i (steady path with K::PRE: all CH loads issued up front)
//   body(c, i, steady, pre)    one sample in a steady / fading chunk
//   fadeAlt(c, lerp, gainOnly) a whole fading chunk, straight-line; returns false to decline
//   gen(c, i, emit)            one sample on the sample-by-sample path, after the state machine has stepped
//   steadyDone(n) / fadeDone(n)  after n samples of a uniform run
//   perChunk()                 after every chunk
// K: compile-time knobs (LoopKnobs).  RUNS: a chunk with an event in it runs [uniform run][event steps][uniform run] -- only
// the event steps need the state machine sample by sample; the run length comes from a bisection with ballots.  (Off for
// the noisy kernels: the extra code costs them more than the runs save, cfg2 15.8 -> 16.8 ms.)  DELAY: lanes start `delay`
// steps late (the lane-pipelined kernel's skew).  STREAM: the launch runs exactly A.maxSamples steps (live handles).
#define STAGE_SYNC() __syncthreads()
template <bool PRE_, bool RUNS_, bool DELAY_, bool STREAM_, bool NOISE_, int UNROLL_, bool LONE_ = false, bool CACHE_ = false>
struct LoopKnobs {
    static constexpr bool PRE = PRE_, RUNS = RUNS_, DELAY = DELAY_, STREAM = STREAM_, NOISE = NOISE_, LONE = LONE_, CACHE = CACHE_;
    static constexpr int UNROLL = UNROLL_;
};

template <class D, int MODE, int CH, class K, class SF, class FForce, class FBegin, class FAltChunk, class FAltSample, class FPre, class FBody, class FFadeAlt,
          class FGen, class FSteadyDone, class FFadeDone, class FChunk>
__device__ __forceinline__ void stage_loop(int depth, int nIter, int nChunks, int fullChunks, int stampSlot, SF& f, PitchState* ps, int32_t* lastIndex,
                                           uint32_t& delay, const int* P, const int* RF, const int* RB, const StageCtx& X,
                                           FForce forceGeneral, FBegin begin, FAltChunk altChunk, FAltSample altSample, FPre preIn, FBody body,
                                           FFadeAlt fadeAlt, FGen gen, FSteadyDone steadyDone, FFadeDone fadeDone, FChunk perChunk)
{
#ifdef KLATT_STAMPS
    Stamps st;
#endif
    constexpr int GI = D::GAIN >= 0 ? D::GAIN : 0;
    // one whole steady chunk (the live lanes)
    auto steadyChunk = [&](int c, bool useAlt) __attribute__((always_inline)) {
        if (!f.done) {
            if (K::PRE) {
                double pre[K::PRE ? CH : 1];
#pragma unroll
                for (int i = 0; i < CH; ++i) pre[i] = preIn(c, i);
                if (useAlt) {
#pragma unroll
                    for (int i = 0; i < CH; ++i) altSample(c, i, pre[i]);
                } else {
#pragma unroll
                    for (int i = 0; i < CH; ++i) body(c, i, true, pre[i]);
                }
            } else if (useAlt) {
                altChunk(c);
            } else {
#pragma unroll K::UNROLL
                for (int i = 0; i < CH; ++i) body(c, i, true, 0.0);
            }
            f.cnt += CH;
            steadyDone(CH);
        }
    };
    // A handle ALONE in its wavefront fills it with 64 identical lanes (streams_synthesize).  What a fade sample costs beyond a steady one --
    // the interpolation of the stage's parameters and exp / cos per moving resonator, 1700 cycles against 280 in the cascade and parallel
    // stages (KLATT_STAMPS build) -- depends on the fade position alone, not on the filter memories: lane i computes it for sample i of
    // the fade's NEXT 64 (a lane past the fade's end repeats the last), all side by side (loneEval), and the recurrence then takes each
    // sample's values from LDS, [slot][sample] (loneTake): one ds_read_b64 of a wave-uniform address hands a value to all lanes (two
    // v_readlane per double and the scalar-operand hazards behind them cost ~450 cycles a sample more).  Same expressions on the same
    // operands as stage_fade, sample for sample.  Slots: a resonator's frequency and bandwidth feed its coefficients alone and have
    // none; where a stretch ends they are interpolated for that one sample, as the state (loneEnd).
    constexpr int NP_ = D::NPARAM > 0 ? D::NPARAM : 1;
    static_assert(!K::LONE || (D::NPARAM + D::NRES + 1) * kLanes * 8 <= kLoneLdsPerStage, "a stage's fade values fit its LDS scratch (every resonator's frequency and bandwidth are parameters of their own, without slots)");
    auto loneOfRes = [&](int k) __attribute__((always_inline)) { bool y = false;
#pragma unroll
        for (int r = 0; r < D::NRES; ++r) y = y || RF[r] == k || RB[r] == k;
        return y; };
    auto loneSlot = [&](int k) __attribute__((always_inline)) { int n = 0;      // parameters below k with a slot of their own
#pragma unroll
        for (int j = 0; j < D::NPARAM; ++j) n += (j < k && !loneOfRes(j)) ? 1 : 0;
        return n; };
    auto loneEval = [&](bool lerp, uint32_t wRes, bool gainOnly, uint32_t coefCls) __attribute__((always_inline)) {
        double* const S = X.loneLds;
        const int coef0 = loneSlot(D::NPARAM), pitchSlot = coef0 + 3 * D::NRES;
        const bool gainAlone = D::GAIN >= 0 && lerp && gainOnly;
        const uint32_t me = threadIdx.x & (kLanes - 1), span = f.newFade - f.cnt;      // (>= 1: the callers are inside the fade)
        const double ratio = div_by((double)(f.cnt + 1u + (me < span ? me : span - 1u)), (double)f.newFade, f.invFade);
        double cv[NP_];
#pragma unroll
        for (int k = 0; k < D::NPARAM; ++k) cv[k] = f.cur[k];
        if (gainAlone) cv[GI] = fade_value(f.oldL[GI * kLanes], f.getNew(GI), ratio);
        else if (lerp) {
#pragma unroll
            for (int k = 0; k < D::NPARAM; ++k) { const double o = f.oldL[k * kLanes], n = f.getNew(k); cv[k] = o + ((n - o) * ratio); }
        }
        if (D::PITCH) S[pitchSlot * kLanes + me] = fade_value(ps->old0, ps->new0, ratio);
        if (gainAlone) S[loneSlot(GI) * kLanes + me] = cv[GI];
        else if (lerp) {
#pragma unroll
            for (int k = 0; k < D::NPARAM; ++k)
                if (!loneOfRes(k)) S[loneSlot(k) * kLanes + me] = cv[k];
        }
#pragma unroll
        for (int r = 0; r < D::NRES; ++r) {
            if (wRes & (1u << r)) {
                const Coef k = D::INLINE_COEF
                    ? resonator_coefficients_inline<MODE>(cv[RF[r]], cv[RB[r]], D::ANTI0 && r == 0, X.A.negPiOverSr, X.A.twoPiOverSr, (int)((coefCls >> (2 * r)) & 3u))
                    : resonator_coefficients<MODE>(cv[RF[r]], cv[RB[r]], D::ANTI0 && r == 0, X.A.negPiOverSr, X.A.twoPiOverSr);
                S[(coef0 + 3 * r) * kLanes + me] = k.a; S[(coef0 + 3 * r + 1) * kLanes + me] = k.b; S[(coef0 + 3 * r + 2) * kLanes + me] = k.c;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    };
    // One evaluation serves the fade's next 64 samples wherever they fall -- the rest of the chunk in which the fade began, whole chunks, the
    // chunk in which it ends: `loneBase` is the sample count the evaluation started from; any step of the state machine ends its validity.
    uint32_t loneBase = 0;
    bool loneValid = false;
    // the values of samples f.cnt + 1 .. f.cnt + n of the running fade are in LDS from here on; returns the first one's place
    auto loneNeed = [&](uint32_t n, bool lerp, uint32_t wRes, bool gainOnly, uint32_t coefCls) __attribute__((always_inline)) -> int {
        if (!(loneValid && f.cnt >= loneBase && f.cnt - loneBase + n <= (uint32_t)kLanes)) {
            loneEval(lerp, wRes, gainOnly, coefCls);
            loneBase = f.cnt; loneValid = true;
        }
        return (int)(f.cnt - loneBase);
    };
    // sample `at` of the evaluation becomes the stage's current values
    auto loneTake = [&](int at, bool lerp, uint32_t wRes, bool gainOnly) __attribute__((always_inline)) {
        const double* const S = X.loneLds;
        const int coef0 = loneSlot(D::NPARAM), pitchSlot = coef0 + 3 * D::NRES;
        if (D::PITCH) ps->cur0 = S[pitchSlot * kLanes + at];
        if (D::GAIN >= 0 && lerp && gainOnly) f.cur[GI] = S[loneSlot(GI) * kLanes + at];
        else if (lerp) {
#pragma unroll
            for (int k = 0; k < D::NPARAM; ++k)
                if (!loneOfRes(k)) f.cur[k] = S[loneSlot(k) * kLanes + at];
        }
#pragma unroll
        for (int r = 0; r < D::NRES; ++r)
            if (wRes & (1u << r)) { f.ra[r] = S[(coef0 + 3 * r) * kLanes + at]; f.rb[r] = S[(coef0 + 3 * r + 1) * kLanes + at]; f.rc[r] = S[(coef0 + 3 * r + 2) * kLanes + at]; }
    };
    // a stretch ends on the sample just taken: the resonators' frequencies and bandwidths (which have no slots) take its values, as the state
    auto loneEnd = [&](bool lerp, bool gainOnly) __attribute__((always_inline)) {
        if (!lerp || (D::GAIN >= 0 && gainOnly) || D::NRES == 0) return;
        const double ratio = div_by((double)f.cnt, (double)f.newFade, f.invFade);
#pragma unroll
        for (int k = 0; k < D::NPARAM; ++k)
            if (loneOfRes(k)) { const double o = f.oldL[k * kLanes], n = f.getNew(k); f.cur[k] = o + ((n - o) * ratio); }
    };
    for (int iter = 0; iter < nIter; ++iter) {
        STAMP_BEGIN();
        STAMP_IDLE();
        int c = iter - depth;
        if (c >= 0 && c < nChunks) {
            const int lim = (K::STREAM && c >= fullChunks) ? (int)(X.A.maxSamples - (uint32_t)c * (uint32_t)CH) : CH;
            int kind = (forceGeneral() || lim < CH) ? -1 : chunk_kind<CH>(f);
            if (kind == 1 && nan_target_live(f)) kind = -1;   // "hold" targets: sample by sample, with the NaN test
            bool lerp = false, gainOnly = false;
            uint32_t wRes = 0;
            if (kind == 1) {
                lerp = __any(!f.done && f.parMask != 0u);
                gainOnly = !K::NOISE && D::GAIN >= 0 && !__any(!f.done && (f.parMask & ~(1u << GI)) != 0u);
                wRes = wave_or_bits<(D::NRES > 0 ? D::NRES : 1)>(f.done ? 0u : f.resMask);
                // a fade in which nothing of THIS element moves is a steady chunk for it (only the counter advances)
                if (!D::PITCH && !lerp && wRes == 0u) kind = 0;
            }
            STAMP_KIND(kind);
            if (kind == 0) {
                const bool useAlt = begin(0);
                // a steady stretch is decided once (steady_run: the minimum over the live lanes of chunks left in it)
                uint32_t run = 1u;
                if (!K::NOISE || KLATT_NOISY_TIGHT) {
                    run = __any(!f.done && f.hasNew) ? 1u : steady_run<CH>(f);   // a fade that moves nothing here: chunk by chunk
                    const uint32_t room = (uint32_t)(fullChunks - c);
                    run = run < room ? run : room;
                }
                for (uint32_t q = 1; q < run; ++q) {
                    steadyChunk(c, useAlt);
                    perChunk();
                    STAMP_WORKED();
                    STAGE_SYNC();
                    STAMP_SYNCED();
                    STAMP_BEGIN();
                    ++iter; ++c;
                }
                steadyChunk(c, useAlt);
            } else if (kind == 1) {
                (void)begin(1);
                const uint32_t coefCls = fade_classes<D>(f, X.A, RF, RB, wRes);   // once per fade stretch
                // what moves in a fade (lerp, wRes, gainOnly) is fixed for the fade: its chunks run in a tight loop too
                auto fadeChunk = [&](int c) __attribute__((always_inline)) {
                    if (K::LONE) {
                        // (an evaluation serves four chunks: loneNeed above)
                        if (!f.done) {
                            const int at = loneNeed((uint32_t)CH, lerp, wRes, gainOnly, coefCls);
#pragma unroll
                            for (int i = 0; i < CH; ++i) {
                                f.cnt++;
                                loneTake(at + i, lerp, wRes, gainOnly);
                                body(c, i, false, 0.0);
                            }
                            loneEnd(lerp, gainOnly);
                            fadeDone(CH);
                        }
                    } else
                    if (!f.done) {
                        if (!fadeAlt(c, lerp, gainOnly)) {
#pragma unroll 2
                            for (int i = 0; i < CH; ++i) {
                                f.cnt++;
                                stage_fade<D, MODE, true>(f, ps, X.A, RF, RB, lerp, wRes, gainOnly, coefCls);
                                body(c, i, false, 0.0);
                            }
                        }
                        fadeDone(CH);
                    }
                };
                uint32_t run = 1u;
                if (KLATT_FADE_TIGHT) {
                    run = fade_run<CH>(f);
                    const uint32_t room = (uint32_t)(fullChunks - c);
                    run = run < room ? run : room;
                }
                for (uint32_t q = 1; q < run; ++q) {
                    fadeChunk(c);
                    perChunk();
                    STAMP_WORKED();
                    STAGE_SYNC();
                    STAMP_SYNCED();
                    STAMP_BEGIN();
                    ++iter; ++c;
                }
                fadeChunk(c);
            } else {
                (void)begin(-1);
                int i = 0;
                // live handles with registers to spare: the fades' end points of the lanes are kept in registers between events (24-28 LDS reads a sample otherwise)
                double co[NP_], cn[NP_];
                bool cached = false;
#pragma nounroll
                while (i < lim) {
                    if (K::LONE && !forceGeneral()) {
                        // a chunk with an event in it, 64 identical lanes: [uniform run][the event's steps][uniform run] -- only the event's
                        // steps need the state machine sample by sample (16 such chunks in a pull of 8192 samples cost a third of it,
                        // 27000 cycles each against 4900 for a steady one); a fading run is evaluated side by side like a fade chunk
                        const bool fad = f.hasNew;
                        const uint32_t rem = f.done ? 0u : (fad ? f.newFade - f.cnt : (f.oldMin > f.cnt ? f.oldMin - f.cnt : 0u));
                        const bool runs = !f.done && !(fad && (f.cnt == 0u || nan_target_live(f)));
                        const uint32_t cap = (uint32_t)(lim - i);
                        const int n = runs ? (int)__builtin_amdgcn_readfirstlane(rem < cap ? rem : cap) : 0;
                        if (n >= 2) {
                            bool lerpR = false, gainOnlyR = false, moves = fad;
                            uint32_t wResR = 0;
                            if (fad) {
                                lerpR = __any(f.parMask != 0u);
                                gainOnlyR = !K::NOISE && D::GAIN >= 0 && !__any((f.parMask & ~(1u << GI)) != 0u);
                                wResR = wave_or_bits<(D::NRES > 0 ? D::NRES : 1)>(f.resMask);
                                if (!D::PITCH && !lerpR && wResR == 0u) moves = false;
                            }
                            if (!moves) {
#pragma nounroll
                                for (int j = i; j < i + n; ++j) body(c, j, true, K::PRE ? preIn(c, j) : 0.0);
                                f.cnt += (uint32_t)n;
                                steadyDone(n);
                            } else {
                                const int at = loneNeed((uint32_t)n, lerpR, wResR, gainOnlyR, kCoefAllUnknown);
#pragma nounroll
                                for (int j = i; j < i + n; ++j) {
                                    f.cnt++;
                                    loneTake(at + (j - i), lerpR, wResR, gainOnlyR);
                                    body(c, j, false, 0.0);
                                }
                                loneEnd(lerpR, gainOnlyR);
                                fadeDone(n);
                            }
                            i += n;
                            continue;
                        }
                    }
                    if (K::RUNS && !K::LONE && !forceGeneral()) {
                        // whenever every live lane is inside a steady stretch (or every one inside a fade, past its first sample, no
                        // NaN target) the next n = min over the lanes of samples left in the stretch run as a rolled loop
                        const bool fad = f.hasNew;
                        const uint32_t rem = f.done ? 0xFFFFFFFFu : (fad ? f.newFade - f.cnt : (f.oldMin > f.cnt ? f.oldMin - f.cnt : 0u));
                        const bool anyFad = __any(!f.done && fad), anySteady = __any(!f.done && !fad);
                        int n = 0;
                        if (!(anyFad && anySteady) && !__any(!f.done && fad && f.cnt == 0u) && !(anyFad && nan_target_live(f))) {
                            const int cap = lim - i;
#pragma unroll
                            for (int st = CH; st >= 1; st >>= 1)
                                if (n + st <= cap && __all(rem >= (uint32_t)(n + st))) n += st;
                        }
                        if (n >= 2) {
                            int kr = anyFad ? 1 : 0;
                            bool lerpR = false, gainOnlyR = false;
                            uint32_t wResR = 0;
                            if (kr == 1) {
                                lerpR = __any(!f.done && f.parMask != 0u);
                                gainOnlyR = D::GAIN >= 0 && !__any(!f.done && (f.parMask & ~(1u << GI)) != 0u);
                                wResR = wave_or_bits<(D::NRES > 0 ? D::NRES : 1)>(f.done ? 0u : f.resMask);
                                if (!D::PITCH && !lerpR && wResR == 0u) kr = 0;
                            }
                            if (!f.done) {
                                if (kr == 0) {
#pragma nounroll
                                    for (int j = i; j < i + n; ++j) body(c, j, true, K::PRE ? preIn(c, j) : 0.0);
                                    f.cnt += (uint32_t)n;
                                    steadyDone(n);
                                } else {
#pragma nounroll
                                    for (int j = i; j < i + n; ++j) {
                                        f.cnt++;
                                        stage_fade<D, MODE, true>(f, ps, X.A, RF, RB, lerpR, wResR, gainOnlyR);
                                        body(c, j, false, 0.0);
                                    }
                                    fadeDone(n);
                                }
                            }
                            i += n;
                            continue;
                        }
                    }
                    if (K::DELAY) {
                        // a lane that has not started yet (pipeline skew) sits this step out
                        const bool hold = delay > 0u;
                        const bool wasDone = f.done;
                        if (hold) { delay--; f.done = true; }
                        const bool emit = stage_advance<D, MODE>(f, ps, lastIndex, P, RF, RB, X);
                        if (hold) f.done = wasDone;
                        gen(c, i, emit);
                    } else if (K::CACHE) {
                        if (!cached) {
#pragma unroll
                            for (int k = 0; k < D::NPARAM; ++k) { co[k] = f.oldL[k * kLanes]; cn[k] = f.getNew(k); }
                            cached = true;
                        }
                        bool anyEvent = false;
                        const bool emit = stage_advance<D, MODE>(f, ps, lastIndex, P, RF, RB, X, co, cn, &anyEvent);
                        if (anyEvent) cached = false;
                        gen(c, i, emit);
                    } else {
                        const bool emit = stage_advance<D, MODE>(f, ps, lastIndex, P, RF, RB, X);
                        if (K::LONE) loneValid = false;
                        gen(c, i, emit);
                    }
                    ++i;
                }
            }
            perChunk();
        }
        STAMP_WORKED();
        STAGE_SYNC();
        STAMP_SYNCED();
    }
#ifdef KLATT_STAMPS
    if (X.A.debug && (threadIdx.x & (kLanes - 1)) == 0) {
        unsigned long long* o = X.A.debug + (blockIdx.x * 4 + stampSlot) * 8;
        o[0] = st.work; o[1] = st.wait; o[2] = st.n[0]; o[3] = st.n[1]; o[4] = st.n[2]; o[5] = st.c[0]; o[6] = st.c[1]; o[7] = st.c[2];
    }
#endif
    (void)stampSlot;
}

// ---- flat stages (FLAT launches) ------------------------------------------------------------------------------------
// With the gains in the tracks as well (entry kinds 14..19, klatt_device.h) a filter stage needs no frame state machine: what
// changes its state is a list of fades at sample positions that follow from the frame durations alone.  Frame k of an
// utterance is dequeued on sample T_k (T_0 = 0, T_k+1 = T_k + max(min_k, fade_k + 1) + 1; reference src/frame.cpp:41-80) and its
// fade's rows apply to samples T_k + 1 .. T_k + fade_k.  So a flat stage keeps per lane the sample on which its next fade starts,
// the rows left of the running one, one offset and stride per entry kind it uses -- no dequeue / fade-end events, no
// interpolation, no end points in LDS.  (The source stage is flat too: what it keeps is the pitch, which glides with the
// utterance's own sample count.)
// USUAL: the entry kinds of the stage (bit e of its list) that usually change in speech -- the mixed chunks are compiled for this
// set and for all kinds.  R: the type the stage keeps its coefficients, memories and gains in.
template <int STAGE_, int NRES_, int NGAIN_, bool ANTI0_, uint32_t USUAL_ = 0, uint32_t USUAL2_ = 0, class R_ = sig_t>
struct FlatDesc {
    using R = R_;
    static constexpr int STAGE = STAGE_;      // which part of a track the stage reads (klatt_device.h)
    static constexpr int NRES = NRES_, NGAIN = NGAIN_, NE = NRES_ + NGAIN_;
    static constexpr bool ANTI0 = ANTI0_;
    static constexpr uint32_t USUAL = USUAL_, USUAL2 = USUAL2_;
};

// ---- hand-overs of the flat stages: one workgroup barrier per chunk ------------------------------------------------------------------
// (Round 4 measured the alternative -- two counters per pipe in LDS, every stage starting a chunk as soon as ITS inputs are there and
// ITS output buffer is free, more than two buffers per pipe: bit-identical and not faster, cfg2 8.59 -> 8.64 ms, profiles/
// r4_handover_variants.txt.  A stage that "waits at the barrier" is not idle hardware: its SIMD runs the other workgroup's wave.  The
// variant lives on as a patch, tools/variants/.)  The loops are written against this interface:
struct BarrierSync {
    __device__ __forceinline__ void begin(int) const {}
    __device__ __forceinline__ void end(int) const { __syncthreads(); }
    __device__ __forceinline__ void idle() const { __syncthreads(); }
};

// ---- flat filter stages, second form (round 3): rows loaded one sample ahead, straight into the coefficients ---------------------
// The sample-by-sample path above pays, on every sample in which ANY lane fades: two ballots and two divergent blocks, the row's
// loads, their full latency (a wait right behind them, ~900 cycles with the tracks beyond the L2) and then the filters, in a
// rolled loop that keeps every sample's operations apart.  Measured with the loads taken out (a timing-only build, tools/variants/): the
// all-different batch 32.6 -> 24.7 ms against 9.2 for the aligned one -- the structure costs more than the latency.  This form
// has ONE kind of chunk besides the steady one:
//   * a lane's fade state is a row counter (`left`) and an entry index + stride per kind; with the track layout of klatt_device.h
//     (header + matrix) the first fade sample is a row like any other, so a fade start is one switch of the indices;
//   * on sample t the rows of sample t + 1 are loaded: for each kind right after its LAST use in sample t, under the mask of the
//     lanes that have a row, straight into the registers the filters read (no staging registers, no copies): the loads have a whole
//     sample's arithmetic (and the SIMD's other wave) to arrive;
//   * a = 1 - b - c is derived where it is used (two operations; for a lane without a new row it reproduces the value it had), so
//     nothing has to wait for a row at the top of a sample; steady chunks keep `a` in registers as before (re-derived once after
//     a mixed stretch);
//   * the filters run unmasked for every lane (a lane past its end computes into its own pipe slots and tile row, which nobody
//     reads: rowCount / produced bound what is flushed), so a sample is straight-line code between the masked loads;
//   * what a fade start needs (track, mask, length, distance to the next start) is ONE 16-byte FlatRef, loaded when the PREVIOUS
//     fade starts.
#ifndef KLATT_MIX_UNROLL
#define KLATT_MIX_UNROLL 4
#endif
template <int E> using KindTag = std::integral_constant<int, E>;
typedef unsigned int flat_u4 __attribute__((ext_vector_type(4)));
typedef unsigned int flat_u2 __attribute__((ext_vector_type(2)));
typedef double flat_d2 __attribute__((ext_vector_type(2)));
template <class FD>
struct FlatState2 {
    static constexpr int NR = FD::NRES > 0 ? FD::NRES : 1;
    using R = typename FD::R;
    R ra[NR], rb[NR], rc[NR], z1[NR], z2[NR];
    R cur[2 * FD::NGAIN > 0 ? 2 * FD::NGAIN : 1];
    uint32_t idx[FD::NE], stride[FD::NE];      // per kind: byte offset of its next entry in the launch's tracks, bytes per row (0: the kind does not move)
    uint32_t startAt, left, next, nFrames, length, produced;      // startAt: the sample the next fade's first row applies to; left: rows of the running fade not yet loaded
    FlatRef nextRef;                            // frame `next`'s fade, loaded ahead
    uint32_t wmask;                             // wave-uniform: bit e -- kind GE[e] changes, in some lane of the wavefront, after the first sample of the first fade
    bool live;
};
template <class R> struct ResPre { R a, p1, p2; };      // a resonator's first half (res_pre below)
template <class T> __device__ __forceinline__ void flat_pin(T v) { asm volatile("" :: "v"(v) : "memory"); }
template <class R> __device__ __forceinline__ void flat_pin(const ResPre<R>& q) { asm volatile("" :: "v"(q.a), "v"(q.p1), "v"(q.p2) : "memory"); }
// OR over the wavefront of the utterances' kind masks (UttDesc.flags), restricted to a stage's kinds: bit e <- kind GE[e]
template <int NE>
__device__ __forceinline__ uint32_t flat2_wave_kinds(uint32_t flags, bool live, const int* GE)
{
    const uint32_t mine = live ? (flags >> kUttKindShift) : 0u;
    uint32_t w = 0;
#pragma unroll
    for (int e = 0; e < NE; ++e) w |= __any((mine >> GE[e]) & 1u) ? (1u << e) : 0u;
    return w;
}
template <class FD>
__device__ __forceinline__ void flat2_init(FlatState2<FD>& f, bool live, const UttDesc& d, const StageCtx& X, const int* GE)
{
#pragma unroll
    for (int r = 0; r < FD::NRES; ++r) { f.ra[r] = 0; f.rb[r] = 2; f.rc[r] = -1; f.z1[r] = 0; f.z2[r] = 0; }
#pragma unroll
    for (int k = 0; k < 2 * FD::NGAIN; ++k) f.cur[k] = 0;
#pragma unroll
    for (int e = 0; e < FD::NE; ++e) { f.idx[e] = 0; f.stride[e] = 0; }
    f.live = live && d.length > 0u;
    f.nFrames = d.nFrames; f.length = d.length; f.next = 0; f.left = 0; f.produced = 0;
    const bool any = live && d.nFrames > 0u;
    f.startAt = any ? 1u : 0xFFFFFFFFu;      // frame 0 is dequeued on sample 0, its fade's first row applies to sample 1
    f.nextRef = FlatRef{0u, 0u, 1u, 0u};
    if (any) f.nextRef = X.myFlat[0];
    f.wmask = (uint32_t)__builtin_amdgcn_readfirstlane((int)flat2_wave_kinds<FD::NE>(d.flags, live, GE));
}
// the fade of frame `next` starts with the coming sample: point every kind at its entries, load the following frame's reference
template <class FD>
__device__ __forceinline__ void flat2_switch(FlatState2<FD>& f, const StageCtx& X, const int* GE)
{
    const FlatRef p = f.nextRef;
    // the stage's part of the track (klatt_device.h): its header (the kinds that do not move), then F rows of those that do
    const uint32_t part = p.off + track_part(p.mask, p.fadeSamples, FD::STAGE);
    const uint32_t nMove = track_stage_slots(p.mask, FD::STAGE), hdr = track_stage_entries(FD::STAGE) - nMove;
    uint32_t moving = 0, still = 0;            // entries of the stage's kinds before this one, among the moving / among the others
#pragma unroll
    for (int e = 0; e < FD::NE; ++e) {
        const bool moves = (p.mask >> GE[e]) & 1u;
        f.idx[e] = (part + (moves ? hdr + moving : still)) * 16u;
        f.stride[e] = moves ? nMove * 16u : 0u;
        const uint32_t w = GE[e] == 0 ? 2u : 1u;      // N0 takes two entries
        moving += moves ? w : 0u; still += moves ? 0u : w;
    }
    f.left = nMove ? p.fadeSamples : 1u;        // a fade that moves nothing of this stage: its first row only
    f.next++;
    const bool more = f.next < f.nFrames;
    f.startAt = more ? f.startAt + p.span : 0xFFFFFFFFu;
    // The following frame's reference, loaded now and read when ITS fade starts.  Everything that reads the current one is pinned
    // before the load (an asm statement with those values as inputs, clobbering memory): its registers are then dead and the load
    // lands in them -- hoisted above, it would land in temporaries and be moved home behind a wait at the end of this block.
#pragma unroll
    for (int e = 0; e < FD::NE; ++e) { flat_pin(f.idx[e]); flat_pin(f.stride[e]); }
    flat_pin(f.left); flat_pin(f.startAt);
    f.nextRef = X.myFlat[more ? f.next : f.nFrames - 1u];      // (past the last frame: any valid frame; never read)
}
// Called by a stage's sample BETWEEN its two halves: the first half has read every coefficient and gain (a = 1 - b - c, the
// products b * z1 and c * z2 -- which depend on the previous samples only --, copies of the gains); here the lanes with a row for the
// NEXT sample load it, all kinds in one masked block, straight into the registers the first half reads; the second half (the
// chains through the sample's input) runs while the loads are in flight.  `firstHalf`: the values of the first half -- empty asm
// statements (inputs: those values; clobber: memory) keep the compiler from hoisting the loads above the first half (it would
// have to copy the old values aside) and from sinking the first half below them.  SET (compile time): the kinds loaded -- a
// chunk runs the instantiation for the stage's USUAL kinds when no lane of the wavefront ever changes another one, else the
// one for all kinds (a wave-uniform test per kind and sample cost more in branches than the loads it saved).
template <class FD, uint32_t SET, bool ALLROWS = false>
struct FlatMid {
    FlatState2<FD>& f;
    __amdgpu_buffer_rsrc_t rsrc;
    bool has;      // (ALLROWS: every lane has a row -- no mask)
    template <class... T>
    __device__ __forceinline__ void operator()(T... firstHalf) const
    {
        using R = typename FD::R;
        (flat_pin(firstHalf), ...);
        if (ALLROWS || has) {
#pragma unroll
            for (int e = 0; e < FD::NE; ++e) {
                if (!(SET & (1u << e))) continue;
                const uint32_t off = f.idx[e];
                const flat_d2 v = __builtin_bit_cast(flat_d2, __builtin_amdgcn_raw_buffer_load_b128(rsrc, off, 0, 0));
                if (e < FD::NRES) {
                    f.rb[e < FD::NRES ? e : 0] = (R)v.x; f.rc[e < FD::NRES ? e : 0] = (R)v.y;
                    if (FD::ANTI0 && e == 0) f.ra[0] = (R)__builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(rsrc, off + 16u, 0, 0));
                } else {
                    const int g = e < FD::NRES ? 0 : e - FD::NRES;
                    f.cur[2 * g] = (R)v.x; f.cur[2 * g + 1] = (R)v.y;
                }
                f.idx[e] = off + f.stride[e];
            }
        }
    }
};
struct NoMid { template <class... T> __device__ __forceinline__ void operator()(T...) const {} };
// A resonator's sample in two halves.  First half (before the rows of the next sample are loaded over b and c): a -- DERIVEd from
// b and c in a chunk that loads the resonator's kind, else the register -- and the two products that do not depend on the
// sample's input.  Second half: ((a * in) + (b * z1)) + (c * z2), the reference's order
// (src/speechWaveGenerator.cpp:131); MODE_FAST fuses, so its first half only sets b and c aside.
template <int MODE, bool DERIVE, class R>
__device__ __forceinline__ ResPre<R> res_pre(R ra, R rb, R rc, R z1, R z2)
{
    ResPre<R> p;
    p.a = DERIVE ? (R)((R)1 - rb - rc) : ra;
    if (MODE == MODE_FAST || !std::is_same<R, double>::value) { p.p1 = rb; p.p2 = rc; }
    else { p.p1 = rb * z1; p.p2 = rc * z2; }
    return p;
}
template <int MODE, class R>
__device__ __forceinline__ R res_post(const ResPre<R>& p, R in, R& z1, R& z2)
{
    R y;
    if (MODE == MODE_FAST || !std::is_same<R, double>::value) y = dot3<MODE>(p.a, in, p.p1, z1, p.p2, z2);
    else y = p.a * in + p.p1 + p.p2;
    z2 = z1; z1 = y;
    return y;
}

// body(c, i, std::bool_constant<MIXED>, gate, mid): one sample of the stage; same barrier discipline as flat_loop
template <class FD, int CH, class FBody, class FChunk, class FSync = BarrierSync>
__device__ __forceinline__ void flat2_loop(int depth, int nIter, int nChunks, int stampSlot, FlatState2<FD>& f, const StageCtx& X, const int* GE, FBody body, FChunk perChunk,
                                           const FSync& sync = FSync{})
{
#ifdef KLATT_STAMPS
    Stamps st;
#endif
    using R = typename FD::R;
    uint32_t staleG = 0;       // wave-uniform: the kinds whose `a` in registers predates the last rows (mixed chunks derive it on the fly)
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<double2*>(X.A.track), 0, (int)X.A.trackBytes, 0x00020000);
    for (int iter = 0; iter < nIter; ++iter) {
        STAMP_BEGIN();
        STAMP_IDLE();
        const int c = iter - depth;
        const bool inRange = c >= 0 && c < nChunks;
        if (inRange) {
            sync.begin(c);
            const uint32_t t0 = (uint32_t)c * (uint32_t)CH, t1 = t0 + (uint32_t)CH;
            if (f.length <= t0) f.live = false;                               // this lane has emitted its last sample
            // a chunk is steady when no lane loads a row in it: none pending, no fade whose first row applies to t0 + 1 .. t1
            const bool busy = f.left > 0u || f.startAt <= t1;
            if (!__any(busy)) {
                if (staleG) {
#pragma unroll
                    for (int r = (FD::ANTI0 ? 1 : 0); r < FD::NRES; ++r)
                        if (staleG & (1u << r)) f.ra[r] = (R)((R)1 - f.rb[r] - f.rc[r]);
                    staleG = 0;
                }
                // decided once: the chunks until some live lane's next fade comes into reach run in a tight loop
                uint32_t run = f.live ? (f.startAt - t0 - 1u) / (uint32_t)CH : 0xFFFFFFFFu;
#pragma unroll
                for (int m = 32; m >= 1; m >>= 1) { const uint32_t o = (uint32_t)__shfl_xor((int)run, m, kLanes); run = o < run ? o : run; }
                run = (uint32_t)__builtin_amdgcn_readfirstlane((int)run);
                const uint32_t room = (uint32_t)(nChunks - c);
                run = run < room ? run : room;
                run = run < 1u ? 1u : run;
                int cc = c;
                STAMP_KIND(0);
                for (uint32_t q = 1; q < run; ++q) {
                    if (f.live) {
#pragma unroll
                        for (int i = 0; i < CH; ++i) body(cc, i, std::integral_constant<uint32_t, 0u>{}, NoMid{});
                    }
                    { const uint32_t e1 = (uint32_t)(cc + 1) * (uint32_t)CH; f.produced = f.length < e1 ? f.length : e1; }
                    perChunk();
                    STAMP_WORKED();
                    sync.end(cc);
                    STAMP_SYNCED();
                    STAMP_BEGIN();
                    ++iter; ++cc;
                    sync.begin(cc);
                }
                if (f.live) {
#pragma unroll
                    for (int i = 0; i < CH; ++i) body(cc, i, std::integral_constant<uint32_t, 0u>{}, NoMid{});
                }
                { const uint32_t e1 = (uint32_t)(cc + 1) * (uint32_t)CH; f.produced = f.length < e1 ? f.length : e1; }
                perChunk();
                STAMP_WORKED();
                sync.end(cc);
                STAMP_SYNCED();
                continue;
            }
            STAMP_KIND(-1);
            // The kinds loaded in this chunk: every one in chunk 0 (the first fade's first row sets them all); afterwards the stage's
            // usual ones when no lane of the wavefront ever changes another kind (f.wmask), else every one again.
            auto mixedChunk = [&](auto setTag, auto allRowsTag) __attribute__((always_inline)) {
                constexpr uint32_t SET = decltype(setTag)::value;
                constexpr bool ALLROWS = decltype(allRowsTag)::value;      // every live lane loads a row on every sample of the chunk, no fade starts: no masks, no look-out
                const uint32_t fix = staleG & ~SET;      // kinds loaded in earlier chunks and not in this one: their `a` goes back to the register
                if (fix) {
#pragma unroll
                    for (int r = (FD::ANTI0 ? 1 : 0); r < FD::NRES; ++r)
                        if (fix & (1u << r)) f.ra[r] = (R)((R)1 - f.rb[r] - f.rc[r]);
                    staleG &= SET;
                }
                staleG |= SET;
#pragma unroll KLATT_MIX_UNROLL
                for (int i = 0; i < CH; ++i) {
                    if (!ALLROWS) {
                        const uint32_t tn = t0 + (uint32_t)i + 1u;            // the sample whose rows this one loads
                        if (tn == f.startAt) flat2_switch<FD>(f, X, GE);
                    }
                    const bool has = ALLROWS || f.left > 0u;
                    body(c, i, setTag, FlatMid<FD, SET, ALLROWS>{f, rsrc, has});
                    if (!ALLROWS && has) f.left--;
                }
                if (ALLROWS) f.left = f.left > (uint32_t)CH ? f.left - (uint32_t)CH : 0u;      // (a lane past its end had none)
            };
            constexpr uint32_t ALL = (1u << FD::NE) - 1u;
            const bool usual = FD::USUAL != 0u && FD::USUAL != ALL && c != 0 && (f.wmask & ~FD::USUAL) == 0u;
            // every live lane inside a fade that moves something of this stage for the whole chunk (time-aligned batches: always, when any)
            const bool allRows = usual && __all(!f.live || (f.left >= (uint32_t)CH && f.startAt > t1));
            if (allRows) mixedChunk(std::integral_constant<uint32_t, FD::USUAL>{}, std::true_type{});
            else if (usual) mixedChunk(std::integral_constant<uint32_t, FD::USUAL>{}, std::false_type{});
            else mixedChunk(std::integral_constant<uint32_t, ALL>{}, std::false_type{});
            f.produced = f.length < t1 ? f.length : t1;
            perChunk();
        }
        STAMP_WORKED();
        if (inRange) sync.end(c); else sync.idle();
        STAMP_SYNCED();
    }
#ifdef KLATT_STAMPS
    if (X.A.debug && (threadIdx.x & (kLanes - 1)) == 0) {
        unsigned long long* o = X.A.debug + (blockIdx.x * 4 + stampSlot) * 8;
        o[0] = st.work; o[1] = st.wait; o[2] = st.n[0]; o[3] = st.n[1]; o[4] = st.n[2]; o[5] = st.c[0]; o[6] = st.c[1]; o[7] = st.c[2];
    }
#endif
    (void)stampSlot;
}

// ---- the kernel ---------------------------------------------------------------------------
// NASAL = false (quiet launches only): every utterance of the launch is nasal-free (UTT_NO_NASAL, classified on the
// host): caNP == 0 in every frame with bounded, stable N0/NP parameters.  The cascade input then passes the nasal pair
// untouched (reference :151-152: x + (np - x) * 0 == x for finite np) and nothing else reads N0's or NP's memories,
// so both are skipped and the six formant resonators are spread evenly:
//   quiet, nasal-free   S0 frame + glottal source | S1 r6, r5, r4 | S2 r3, r2, r1 | S3 gain, clip, int16 -> PCM
// STREAM (noisy launches only): the lanes are LIVE handles (speechPlayer_synthesizeMany): every stage restores its slice of
// the handle's saved state, applies a pending purge, the launch advances every handle by exactly A.maxSamples steps (a
// handle whose queue runs dry stops earlier), and every stage saves its slice again.  Live handles always take the
// noisy instantiation: the noise generators' memories and counters advance with every sample whatever the gains
// (reference src/speechWaveGenerator.cpp:39-42), and a handle that is quiet now may be given noisy frames later.
// FLAT (noisy batch launches only): the utterances of the launch have tracks (klatt_tracks.h; host: UTT_TRACKED) and all four
// stages are flat stages (above): no exp / cos, no interpolation, no frame state machine in the sample loop.
// LONE: the launch is ONE workgroup whose 64 lanes advance the same live handle (streams_synthesize's replicas): an instantiation of its
// own, so that the fade chunks' side-by-side path (stage_loop) costs the other live-handle kernels neither registers nor code
template <int MODE, bool NOISE, int CH, int WPS = KLATT_MINWAVES, bool NASAL = true, bool STREAM = false, bool FLAT = false, bool LONE = false>
__global__ void __launch_bounds__(kLanes * kStages, WPS) klatt_systolic(const KernelArgs A)
{
    static_assert(!LONE || STREAM, "LONE: a live handle pulled alone");
    static_assert(NASAL || !NOISE, "only quiet launches have a nasal-free instantiation");
    static_assert(NOISE || !STREAM, "live handles take the noisy instantiation");
    static_assert(!FLAT || (NOISE && !STREAM), "tracks: noisy batch launches");
    constexpr int SPLIT = stream_split<STREAM, WPS, LONE>();
    constexpr int MOVED = SPLIT & 3;
    using L = SysLds<NOISE, CH, FLAT, SPLIT>;
    static_assert(SPLIT != 3, "eight resonators in the head stage: not built");
    constexpr int kChunk = CH;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    using PipeT = typename std::conditional<FLAT, sig_t, double>::type;                // what the stages hand over
    PipeT* const pipeX = reinterpret_cast<PipeT*>(lds);                             // S0 -> S1
    PipeT* const pipeO = reinterpret_cast<PipeT*>(lds + L::kPipeO);                 // S1 -> S2
    PipeT* const pipeA = reinterpret_cast<PipeT*>(lds + L::kPipeO + L::kPipeBytes); // noisy: y      | quiet: S2 -> S3
    PipeT* const pipeB = reinterpret_cast<PipeT*>(lds + L::kPipeO + (NOISE ? 2 : 1) * L::kPipeBytes);     // noisy: partial sum
    constexpr int nbuf_pipeX = L::kBufsX, nbuf_pipeO = L::kBufs, nbuf_pipeA = L::kBufs, nbuf_pipeB = L::kBufs;
    (void)nbuf_pipeX; (void)nbuf_pipeO; (void)nbuf_pipeA; (void)nbuf_pipeB;
    unsigned char* const tile = lds + L::kTileOff;
    long long* const rowBase = reinterpret_cast<long long*>(lds + L::kRowBase);
    uint32_t* const rowCount = reinterpret_cast<uint32_t*>(lds + L::kRowCount);
    uint32_t* const maxLenP = reinterpret_cast<uint32_t*>(lds + L::kMaxLen);

    const int lane = threadIdx.x & (kLanes - 1);
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const long long slot = (long long)blockIdx.x * kLanes + lane;
    const uint32_t u = (slot < A.nSlots) ? A.order[slot] : 0xFFFFFFFFu;
    const bool live = (u != 0xFFFFFFFFu);

    UttDesc d;
    d.frameStart = 0; d.outStart = 0; d.nFrames = 0; d.seed = 0; d.flags = 0; d.length = 0;
    if (live) d = A.utt[u];
    const FrameWindow fw = frame_window(A, d);
    const StageCtx X{A, d, A.frames + fw.base * kNumParams, A.meta + fw.base, fw.off, fw.mask,
                     FLAT ? A.flatRef + d.frameStart : nullptr, FLAT ? A.sourceRef + d.frameStart : nullptr, LONE,
                     LONE ? reinterpret_cast<double*>(lds + L::kBytes + wave * kLoneLdsPerStage) : nullptr};
    const uint32_t nkey = noise_key(d.seed), ninc = noise_inc(d.seed), ninc2 = noise_inc2(ninc);
    constexpr int FINAL = NOISE ? 2 : 3;
    // quiet launches read a steady chunk's inputs from the pipe up front (the loads of all CH samples go out
    // together instead of one LDS latency per unrolled group); the noisy kernels have no registers to spare
#ifndef KLATT_PRELOAD
#define KLATT_PRELOAD 1
#endif
    constexpr bool kPre = !NOISE && KLATT_PRELOAD;

    // Which wave runs which stage.  With two workgroups per CU every SIMD hosts one wave of each, and the
    // hardware places wave w of the newer workgroup beside wave w + 1 of the older one (tools/hwid2.hip), which
    // puts the two heaviest noisy stages on one SIMD.  Choosing the stage from the SIMD the wave sits on and the
    // parity of its wave slot pairs S0 with S2 and S1 with S3 instead (noisy: 34 + 16 work units per SIMD pair become
    // 25 + 25; quiet nasal-free, in f64 operations per sample: 27 + 30 + 24 + 21 become 27 + 24).  The choice is only taken when it gives the
    // four waves four different stages (checked through LDS); otherwise stage = wave index.  Any bijection is
    // correct -- the waves are interchangeable until they pick a stage.
    uint32_t* const stageMaskP = maxLenP + 1;
    uint32_t* const syncP = reinterpret_cast<uint32_t*>(lds + L::kSync);
    if (threadIdx.x == 0) { *maxLenP = 0; *stageMaskP = 0; }
    if (threadIdx.x < 8) syncP[threadIdx.x] = 0;
    __syncthreads();
    int cand = wave;
    if (KLATT_PAIR && WPS == 2 && (NOISE || !NASAL)) {
        const uint32_t hw = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));   // HW_REG_HW_ID: wave slot [3:0], SIMD [5:4]
        cand = (int)((((hw >> 4) & 3u) + ((hw & 1u) ? 2u : 0u)) & 3u);
    }
    if (wave == 0) atomicMax(maxLenP, d.length);
    if (lane == 0) atomicOr(stageMaskP, 1u << cand);
    __syncthreads();
    const int stage = __builtin_amdgcn_readfirstlane((*stageMaskP == 0xFu) ? cand : wave);
    if (stage == FINAL) { rowBase[lane] = d.outStart; rowCount[lane] = 0; }   // read by this wave only
    const uint32_t maxLen = *maxLenP;
    const int nChunks = (int)((maxLen + kChunk - 1) / kChunk);
    const int nIter = nChunks + (NOISE ? 2 : 3);   // the final stage lags 2 (noisy) or 3 (quiet) chunks; same trip count in every wave
    // Live handles: every handle that has frames left emits one sample per step, so "at most maxSamples samples per handle" is
    // "exactly maxSamples steps" for the whole launch (host: UttDesc.length = maxSamples); the last chunk may be a partial one
    const int fullChunks = STREAM ? (int)(A.maxSamples / (uint32_t)kChunk) : nChunks;
    double* const streamState = (STREAM && live) ? (A.statePtrs ? A.statePtrs[u] : A.state + (size_t)u * kStateDoubles) : nullptr;
    const bool streamPurge = STREAM && live && A.control && (A.control[u] & 1u);
    // pipe slot of sample i of chunk c
#define PIPE(p, c, i) (p)[(((nbuf_##p) == 2 ? ((c) & 1) : ((c) % (nbuf_##p))) * kChunk + (i)) * kLanes + lane]
    // the chunk loop's knobs: quiet launches preload a steady chunk's inputs and run uniform stretches inside event chunks
    using KSrc = LoopKnobs<false, (!NOISE || KLATT_NOISY_RUNS), false, STREAM, NOISE, KLATT_UNROLL, LONE, (STREAM && WPS == 1 && !LONE)>;      // stages without a pipe input
    using KFil = LoopKnobs<kPre, (!NOISE || KLATT_NOISY_RUNS), false, STREAM, NOISE, KLATT_UNROLL, LONE, (STREAM && WPS == 1 && !LONE)>;       // stages that read a pipe
    uint32_t noDelay = 0;
    auto never = [&]() __attribute__((always_inline)) { return false; };
    auto noBegin = [&](int) __attribute__((always_inline)) { return false; };
    auto noAlt = [&](int) __attribute__((always_inline)) {};
    auto noAltSample = [&](int, int, double) __attribute__((always_inline)) {};
    auto noFadeAlt = [&](int, bool, bool) __attribute__((always_inline)) { return false; };
    auto nothing = [&](int) __attribute__((always_inline)) {};
    auto noChunk = [&]() __attribute__((always_inline)) {};

    // hand-overs of the flat stages (BarrierSync above): pipe X = S0 -> S1, O = S1 -> final, A and B = S3 -> final
    auto make_sync = [&](auto, auto, auto, int, int, int) __attribute__((always_inline)) { return BarrierSync{}; };
    using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>; using I2 = std::integral_constant<int, 2>;
    if (FLAT && KLATT_FLAT_SOURCE && stage == 0) {
        // ================= flat S0, second form: the source stage with its rows loaded one sample ahead (see flat2_loop) =================
        // Its seven parameters come from the tracks (entry kinds 20..23); what stays here is the pitch, which glides with the sample
        // count of THIS utterance (reference src/frame.cpp:76-79, :98, :71) and so cannot be shared.  Frame k is dequeued on sample
        // T_k (the sample before its fade's first row: the stage's `startAt - 1`); per sample a lane is dequeuing (sets up the pitch
        // fade and the rows; the sample itself is emitted unchanged), fading (pitch interpolated), ending its fade (bookkeeping) or
        // steady (glide).  What a dequeue reads (durations, index mark, the frame's two pitch values, the FlatRef) was loaded when the
        // previous frame was dequeued.
        if constexpr (FLAT) {
#if KLATT_FLAT_LAYOUT == 2
            // the source AND the head of the cascade: N0 (anti), NP mixed in by caNP (reference src/speechWaveGenerator.cpp:149-152)
            using FD = FlatDesc<0, 2, 5, true, 0, 0, double>;
            constexpr int GE[7] = {0, 1, 14, 20, 21, 22, 23};     // N0, NP | cur: caNP, - | vibratoPitchOffset, vibratoSpeed, turbulence, openQuotient, voiceAmplitude, aspirationAmplitude, preFormantGain
            constexpr int CB = 2;                                 // f.cur[CB + k]: the source's parameters
            constexpr uint32_t kUsualS0 = 0x67u, kAllS0 = 0x7Fu, kVibBit = 8u;      // usually: N0, NP, caNP, the amplitudes and the gain
            constexpr bool kHead = true;
#else
            using FD = FlatDesc<0, 0, 4, false, 0, 0, double>;
            constexpr int GE[4] = {20, 21, 22, 23};     // cur: vibratoPitchOffset, vibratoSpeed, turbulence, openQuotient, voiceAmplitude, aspirationAmplitude, preFormantGain
            constexpr int CB = 0;
            constexpr uint32_t kUsualS0 = 0xCu, kAllS0 = 0xFu, kVibBit = 1u;
            constexpr bool kHead = false;
#endif
            bool staleNP = false;       // NP's `a` in its register predates the last rows (mixed chunks derive it on the fly)
            FlatState2<FD> f;
            flat2_init<FD>(f, live, d, X, GE);
            PitchState ps;
            ps.cur0 = 0.0; ps.old0 = 0.0; ps.new0 = 0.0; ps.oldInc = 0.0; ps.newInc = 0.0;
            double pitchPhase = 0.0, vibPhase = 0.0, aspNoise = 0.0, invFade = 1.0, nfD = 1.0;
            uint32_t noiseSt = noise_first(nkey, ninc), cntF = 0, nfU = 0;    // noiseSt: the state of this stage's next noise value (aspiration: values 0, 2, 4, ...); cntF of nfU pitch-fade samples done
            uint32_t fadeEndAt = 0xFFFFFFFFu;
            int32_t lastIndex = -1;
            bool oldNull = true, newNull = false;
            SourceRef nextSrc{0.0, 0.0, 1.0, -1, 0u};     // frame `f.next`, loaded ahead like f.nextRef
            if (live && d.nFrames > 0u) nextSrc = X.mySrc[0];
            auto source = [&](bool waveVib, auto setTag, const auto& mid) __attribute__((always_inline)) -> double {
                constexpr uint32_t SET = decltype(setTag)::value;
                double vib = 1.0;
                if (waveVib) {
                    const double vs = f.cur[CB + 1];
                    const double adv = frac_toward_zero(div_by(vs, A.sampleRateF, A.invSampleRate) + vibPhase);
                    vibPhase = (vs != 0.0) ? adv : vibPhase;
                    vib = (sin(vibPhase * 6.283185307179586) * 0.06 * f.cur[CB + 0]) + 1.0;
                }
                const double turbGain = f.cur[CB + 2], openQ = f.cur[CB + 3], voiceAmp = f.cur[CB + 4], aspAmp = f.cur[CB + 5], preGain = f.cur[CB + 6];
                pitchPhase = frac_toward_zero(div_by(ps.cur0 * vib, A.sampleRateF, A.invSampleRate) + pitchPhase);
                double voice = (pitchPhase * 2.0) - 1.0;
                aspNoise = noise_uniform(noiseSt) + 0.75 * aspNoise;
                noiseSt = noise_step2(noiseSt, ninc2);
                double asp = aspNoise * 0.2;
                double turb = asp * turbGain;
                turb = (pitchPhase >= openQ) ? turb : turb * 0.01;
                voice += turb;
                voice *= voiceAmp;
                asp *= aspAmp;
                const double src = asp + voice;
                const double out = (src * preGain) * 0.5;
                if constexpr (kHead) {
                    const ResPre<double> q0 = res_pre<MODE, false>(f.ra[0], f.rb[0], f.rc[0], f.z1[0], f.z2[0]);      // N0's a comes from the track
                    const ResPre<double> q1 = res_pre<MODE, ((SET >> 1) & 1u) != 0u>(f.ra[1], f.rb[1], f.rc[1], f.z1[1], f.z2[1]);
                    const double caNP = f.cur[0];
                    mid(out, q0, q1, caNP);      // the next sample's rows (flat2_loop): every parameter has been read
                    double zin = f.z1[0];
                    const double n0 = res_post<MODE>(q0, out, zin, f.z2[0]);       // the anti-resonator remembers its INPUT (:133)
                    f.z1[0] = out;
                    const double np = res_post<MODE>(q1, n0, f.z1[1], f.z2[1]);
                    return fade_value(out, np, caNP);
                } else {
                    mid(out);            // the next sample's rows (flat2_loop): every parameter has been read; they land while the next sample's pitch, phase and noise are computed
                    return out;
                }
            };
            auto vib_live = [&]() __attribute__((always_inline)) -> bool { return f.cur[CB + 0] != 0.0 || f.cur[CB + 1] != 0.0 || vibPhase != vibPhase; };
            const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<double2*>(A.track), 0, (int)A.trackBytes, 0x00020000);
#ifdef KLATT_STAMPS
            Stamps st;
#endif
            const auto sync = make_sync(std::integral_constant<int, L::kBufsX>{}, I0{}, I1{}, 0, 0, 0);
            for (int iter = 0; iter < nIter; ++iter) {
                STAMP_BEGIN();
                STAMP_IDLE();
                const int c = iter;
                int lastChunk = -1;      // the chunk this iteration ends on (the steady stretch below runs several)
                if (c < nChunks) {
                    sync.begin(c);
                    lastChunk = c;
                    const uint32_t t0 = (uint32_t)c * (uint32_t)CH, t1 = t0 + (uint32_t)CH;
                    if (f.length <= t0) f.live = false;
                    const bool busy = f.left > 0u || f.startAt <= t1 || cntF < nfU || fadeEndAt < t1 || vib_live();
                    STAMP_KIND(__any(busy) ? -1 : 0);
                    if (!__any(busy)) {
                        // steady stretch, decided once (as in flat2_loop): the pitch glides, nothing else changes
                        uint32_t run = f.live ? (f.startAt - t0 - 1u) / (uint32_t)CH : 0xFFFFFFFFu;
#pragma unroll
                        for (int m = 32; m >= 1; m >>= 1) { const uint32_t o = (uint32_t)__shfl_xor((int)run, m, kLanes); run = o < run ? o : run; }
                        run = (uint32_t)__builtin_amdgcn_readfirstlane((int)run);
                        const uint32_t room = (uint32_t)(nChunks - c);
                        run = run < room ? run : room;
                        run = run < 1u ? 1u : run;
                        int cc = c;
                        if (kHead && staleNP) { f.ra[1] = 1.0 - f.rb[1] - f.rc[1]; staleNP = false; }
                        for (uint32_t q = 0; q < run; ++q) {
                            if (f.live) {
#pragma unroll
                                for (int i = 0; i < CH; ++i) { ps.cur0 += ps.oldInc; PIPE(pipeX, cc, i) = source(false, std::integral_constant<uint32_t, 0u>{}, NoMid{}); }
                                ps.old0 = ps.cur0;
                            }
                            if (q + 1 < run) { STAMP_WORKED(); sync.end(cc); STAMP_SYNCED(); STAMP_BEGIN(); ++iter; ++cc; sync.begin(cc); lastChunk = cc; }
                        }
                    } else {
                        // the kinds loaded in this chunk (as in flat2_loop): all in chunk 0, afterwards the usual two -- the amplitudes and
                        // the gain -- when no lane of the wavefront ever changes vibrato, turbulence or the open quotient
                        auto mixedChunk = [&](auto setTag, auto allFadeTag) __attribute__((always_inline)) {
                        constexpr uint32_t SET = decltype(setTag)::value;
                        constexpr bool ALLFADE = decltype(allFadeTag)::value;      // every live lane's pitch fades through the whole chunk, nobody dequeues: no selects, no look-out
                        // vibrato can only come alive in this chunk through a row of its kind (the phase only turns NaN while it advances)
                        const bool vibChunk = (SET & kVibBit) != 0u || __any(vib_live());
                        staleNP = true;
#pragma unroll KLATT_MIX_UNROLL
                        for (int i = 0; i < CH; ++i) {
                            const uint32_t t = t0 + (uint32_t)i;
                            const bool deq = !ALLFADE && t + 1u == f.startAt;
                            if (!ALLFADE && __any(deq)) {
                                if (deq) {   // reference src/frame.cpp:55-72 (stage_event restates it); the sample itself is emitted as it is
                                    const SourceRef m = nextSrc;
                                    const uint32_t nf = f.nextRef.fadeSamples;
                                    newNull = (m.flags & FRAME_NULL) != 0;
                                    ps.new0 = newNull ? ps.cur0 : m.pitch;
                                    ps.newInc = newNull ? 0.0 : m.pitchInc;                // reference src/frame.cpp:98 (the division: host)
                                    if (!newNull && oldNull) ps.old0 = m.pitch;
                                    oldNull = newNull;                                    // for the NEXT dequeue: this fade has ended by then (:44-47)
                                    if (m.userIndex != -1) lastIndex = m.userIndex;       // (:69)
                                    nfD = (double)nf; nfU = nf;
                                    ps.new0 += ps.newInc * nfD;                           // (:71)
                                    invFade = m.invFade;
                                    cntF = 0;
                                    fadeEndAt = t + nf + 1u;
                                    flat_pin(ps.new0); flat_pin(ps.newInc); flat_pin(ps.old0); flat_pin(invFade); flat_pin(lastIndex); flat_pin(oldNull);
                                    flat2_switch<FD>(f, X, GE);                           // the rows: the first one applies to sample t + 1
                                    nextSrc = X.mySrc[f.next < f.nFrames ? f.next : f.nFrames - 1u];
                                }
                            }
                            // the pitch of this sample, as selects (every lane computes both candidates: cheaper than three masked blocks):
                            // fading -> interpolated; the sample after the fade -> the fade's target becomes the glide's start; steady -> glide
                            // (reference src/frame.cpp:48-53, :44-47, :76-79); a dequeuing lane leaves it alone
                            if (ALLFADE) {
                                cntF++;
                                ps.cur0 = fade_value(ps.old0, ps.new0, div_by((double)cntF, nfD, invFade));
                            } else {
                            const bool fad = !deq && cntF < nfU;
                            const bool ending = !deq && !fad && t == fadeEndAt;
                            const bool glide = !deq && !fad && !ending;
                            const uint32_t cn = cntF + 1u;
                            const double ratio = div_by((double)cn, nfD, invFade);
                            const double fv = fade_value(ps.old0, ps.new0, ratio);
                            const double gv = ps.cur0 + ps.oldInc;
                            ps.cur0 = fad ? fv : (glide ? gv : ps.cur0);
                            ps.old0 = ending ? ps.new0 : (glide ? gv : ps.old0);
                            ps.oldInc = ending ? ps.newInc : ps.oldInc;
                            cntF = fad ? cn : cntF;
                            fadeEndAt = ending ? 0xFFFFFFFFu : fadeEndAt;
                            }
                            const bool waveVib = vibChunk && __any(vib_live());
                            const bool has = f.left > 0u;
                            PIPE(pipeX, c, i) = source(waveVib, setTag, FlatMid<FD, SET>{f, rsrc, has});
                            if (has) f.left--;
                        }
                        };
                        const bool usual = c != 0 && (f.wmask & ~kUsualS0) == 0u;
                        // (a lane past its end: cntF == nfU; its pitch is nobody's business)
                        const bool allFade = usual && !__any(vib_live()) && __all(!f.live || (cntF + (uint32_t)CH <= nfU && f.startAt > t1));
                        if (allFade) mixedChunk(std::integral_constant<uint32_t, kUsualS0>{}, std::true_type{});
                        else if (usual) mixedChunk(std::integral_constant<uint32_t, kUsualS0>{}, std::false_type{});
                        else mixedChunk(std::integral_constant<uint32_t, kAllS0>{}, std::false_type{});
                    }
                }
                STAMP_WORKED();
                if (lastChunk >= 0) sync.end(lastChunk); else sync.idle();
                STAMP_SYNCED();
            }
#ifdef KLATT_STAMPS
            if (A.debug && lane == 0) {
                unsigned long long* o = A.debug + (blockIdx.x * 4 + 0) * 8;
                o[0] = st.work; o[1] = st.wait; o[2] = st.n[0]; o[3] = st.n[1]; o[4] = st.n[2]; o[5] = st.c[0]; o[6] = st.c[1]; o[7] = st.c[2];
            }
#endif
            if (live) {
                UttResult res;
                res.produced = d.length; res.framesTaken = f.next; res.lastIndex = lastIndex; res.drained = 1u;
                A.result[u] = res;
            }
        }
    } else if (stage == 0) {
        // ================= S0: frame + glottal source (+ aspiration noise) =================
        // tracked: 1 vibratoPitchOffset, 2 vibratoSpeed, 3 turbulence, 4 openQuotient, 5 voiceAmplitude,
        //          6 aspirationAmplitude, 44 preFormantGain (quiet launches never read 3, 4, 6)
        // SPLIT bit 2 (live handles, one workgroup per CU): and the nasal pair N0 (anti), NP mixed in by caNP, behind the source
        constexpr bool NP0 = (SPLIT >> 2) != 0;
        using D = StageDesc<NP0 ? 12 : 7, NP0 ? 2 : 0, 6, true, NP0, NP0>;
        constexpr int P[12] = {1, 2, 3, 4, 5, 6, 44, 13, 21, 14, 22, 23};
        constexpr int RF[2] = {7, 9}, RB[2] = {8, 10};
        StageFrame<D::NPARAM, D::NRES, true> f;         // the target values in registers: S0 has the room, the workgroup's LDS does not
        PitchState ps;
        stage_frame_init(f, live, lds + L::kFrames, lane);
        ps.cur0 = 0.0; ps.old0 = 0.0; ps.new0 = 0.0; ps.oldInc = 0.0; ps.newInc = 0.0;
        double pitchPhase = 0.0, vibPhase = 0.0, aspNoise = 0.0;
        uint32_t noiseSt = noise_first(nkey, ninc);     // the state of this stage's next noise value (aspiration: values 0, 2, 4, ...)
        int32_t lastIndex = -1;
        bool vibFrames = false;
        constexpr int GR0[2] = {0, 1};
        if (STREAM && live) {
            if (streamState[239] != 0.0) {
                pitchPhase = streamState[208]; vibPhase = streamState[209]; aspNoise = streamState[210];
                lastIndex = (int32_t)streamState[220]; noiseSt = (uint32_t)streamState[221];
            }
            stage_state_load<D>(f, &ps, streamState, P, RF, RB, GR0, streamPurge);
            vibFrames = f.oldL[0] != 0.0 || f.oldL[kLanes] != 0.0 || f.getNew(0) != 0.0 || f.getNew(1) != 0.0;
        }

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
                aspNoise = noise_uniform(noiseSt) + 0.75 * aspNoise;
                noiseSt = noise_step2(noiseSt, ninc2);
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
            const double x = (src * f.cur[6]) * 0.5;
            if (!NP0) return x;
            const double n0 = dot3<MODE>(f.ra[0], x, f.rb[0], f.z1[0], f.rc[0], f.z2[0]);
            f.z2[0] = f.z1[0]; f.z1[0] = x;                       // anti-resonator remembers its INPUT (:133)
            const double np = resonate<MODE>(f.z1[NP0 ? 1 : 0], f.z2[NP0 ? 1 : 0], f.ra[NP0 ? 1 : 0], f.rb[NP0 ? 1 : 0], f.rc[NP0 ? 1 : 0], n0);
            return fade_value(x, np, f.cur[NP0 ? 11 : 0]);
        };
        auto vib_live_now = [&]() __attribute__((always_inline)) -> bool {
            return vibFrames || f.cur[0] != 0.0 || f.cur[1] != 0.0 || vibPhase != vibPhase;
        };
        stage_loop<D, MODE, CH, KSrc>(0, nIter, nChunks, fullChunks, stage, f, &ps, &lastIndex, noDelay, P, RF, RB, X,
            [&]() __attribute__((always_inline)) { return __any(!f.done && vib_live_now()); },
            // Straight-line chunks of the quiet source (vibrato off, checked above).  Steady chunk without a pitch glide in any
            // live lane: cur0 + 0 repeated is cur0 + 0 once, and the phase increment (cur0 * 1) / sr is one value for the chunk.
            [&](int kind) __attribute__((always_inline)) -> bool { return !NOISE && kind == 0 && !__any(!f.done && ps.oldInc != 0.0); },
            [&](int c) __attribute__((always_inline)) {
                ps.cur0 += ps.oldInc;
                const double inc = div_by(ps.cur0 * 1.0, A.sampleRateF, A.invSampleRate);
                // the phase recurrence (three dependent operations per sample) first, then the element-wise rest, whose
                // operations are independent across samples and fill the recurrence's issue gaps
                double ph[kChunk];
#pragma unroll
                for (int i = 0; i < kChunk; ++i) { pitchPhase = frac_toward_zero(inc + pitchPhase); ph[i] = pitchPhase; }
#pragma unroll
                for (int i = 0; i < kChunk; ++i) PIPE(pipeX, c, i) = ((((ph[i] * 2.0) - 1.0) * f.cur[4]) * f.cur[6]) * 0.5;
            },
            noAltSample,
            [&](int, int) __attribute__((always_inline)) { return 0.0; },
            [&](int c, int i, bool steady, double) __attribute__((always_inline)) {
                if (steady) ps.cur0 += ps.oldInc;
                PIPE(pipeX, c, i) = source(false);
            },
            // Fading chunk in which only the gain (and the pitch) move -- fades into and out of silence, reference
            // src/frame.cpp:59-67 -- and no target is NaN ("hold", src/utils.h:21): from + ((to - from) * ratio) with the
            // differences taken once, the old/new values read from LDS once.  Same operations on the same operands as
            // stage_fade + source, sample by sample.
            [&](int c, bool lerp, bool gainOnly) __attribute__((always_inline)) -> bool {
                if (NOISE || !gainOnly) return false;
                constexpr int GI = 6;
                const double g0 = lerp ? f.oldL[GI * kLanes] : f.cur[GI], g1 = lerp ? f.getNew(GI) : f.cur[GI];
                if (__any(g1 != g1 || ps.new0 != ps.new0)) return false;
                const double gd = g1 - g0, p0 = ps.old0, pd = ps.new0 - p0, nf = (double)f.newFade;
                if (!__any(pd != 0.0 || p0 != p0)) {
                    // the pitch does not move either: p0 + ((p0 - p0) * ratio) == p0 + 0
                    ps.cur0 = p0 + 0.0;
                    const double inc = div_by(ps.cur0 * 1.0, A.sampleRateF, A.invSampleRate);
#pragma unroll
                    for (int i = 0; i < kChunk; ++i) {
                        f.cnt++;
                        const double ratio = div_by((double)f.cnt, nf, f.invFade);
                        const double gain = g0 + (gd * ratio);
                        pitchPhase = frac_toward_zero(inc + pitchPhase);
                        PIPE(pipeX, c, i) = ((((pitchPhase * 2.0) - 1.0) * f.cur[4]) * gain) * 0.5;
                        if (i == kChunk - 1) f.cur[GI] = gain;
                    }
                } else {
#pragma unroll
                    for (int i = 0; i < kChunk; ++i) {
                        f.cnt++;
                        const double ratio = div_by((double)f.cnt, nf, f.invFade);
                        ps.cur0 = p0 + (pd * ratio);
                        const double gain = g0 + (gd * ratio);
                        pitchPhase = frac_toward_zero(div_by(ps.cur0 * 1.0, A.sampleRateF, A.invSampleRate) + pitchPhase);
                        PIPE(pipeX, c, i) = ((((pitchPhase * 2.0) - 1.0) * f.cur[4]) * gain) * 0.5;
                        if (i == kChunk - 1) f.cur[GI] = gain;
                    }
                }
                return true;
            },
            [&](int c, int i, bool emit) __attribute__((always_inline)) {
                if (emit && f.hasNew && f.cnt == 0)
                    vibFrames = f.oldL[0] != 0.0 || f.oldL[kLanes] != 0.0 || f.getNew(0) != 0.0 || f.getNew(1) != 0.0;
                const bool waveVib = __any(emit && vib_live_now());
                if (emit) { PIPE(pipeX, c, i) = source(waveVib); f.produced++; }
            },
            [&](int n) __attribute__((always_inline)) { ps.old0 = ps.cur0; f.produced += n; },
            [&](int n) __attribute__((always_inline)) { f.produced += n; },
            noChunk);
        if (live) {
            UttResult res;
            res.produced = f.produced; res.framesTaken = f.nextFrame; res.lastIndex = lastIndex; res.drained = STREAM ? (f.done ? 1u : 0u) : 1u;
            A.result[u] = res;
        }
        if (STREAM && live) {
            stage_state_save<D>(f, &ps, streamState, P, GR0);
            streamState[208] = pitchPhase; streamState[209] = vibPhase; streamState[210] = aspNoise;
            streamState[220] = (double)lastIndex; streamState[221] = (double)noiseSt;
        }
    } else if (FLAT && stage == 1) {
        // ================= flat S1: layout 2: the cascade r6 .. r1; layout 1: N0 (anti), NP mixed by caNP, r6 .. r3 =================
        if constexpr (FLAT) {
#if KLATT_FLAT_LAYOUT == 2
            using FD = FlatDesc<1, 6, 0, false, 0x38u>;                // usually r3, r2, r1, when anything
            constexpr int GE[6] = {2, 3, 4, 5, 6, 7};            // r6, r5, r4, r3, r2, r1
            FlatState2<FD> f;
            flat2_init<FD>(f, live, d, X, GE);
            flat2_loop<FD, CH>(1, nIter, nChunks, stage, f, X, GE,
                [&](int c, int i, auto setTag, const auto& mid) __attribute__((always_inline)) {
                    constexpr uint32_t SET = decltype(setTag)::value;      // the kinds this chunk loads (0: a steady chunk)
                    ResPre<sig_t> q[6];
#pragma unroll
                    for (int r = 0; r < 6; ++r) q[r] = ((SET >> r) & 1u) ? res_pre<MODE, true>(f.ra[r], f.rb[r], f.rc[r], f.z1[r], f.z2[r])
                                                                         : res_pre<MODE, false>(f.ra[r], f.rb[r], f.rc[r], f.z1[r], f.z2[r]);
                    mid(q[0], q[1], q[2], q[3], q[4], q[5]);
                    sig_t o = PIPE(pipeX, c, i);
#pragma unroll
                    for (int r = 0; r < 6; ++r) o = res_post<MODE>(q[r], o, f.z1[r], f.z2[r]);
                    PIPE(pipeO, c, i) = o;
                },
                noChunk, make_sync(std::integral_constant<int, L::kBufs>{}, I1{}, I1{}, 0, 0, 2));
#else
            using FD = FlatDesc<1, 6, 1, true, 0x63u>;                 // usually N0, NP, r3 and caNP, when anything
            constexpr int GE[7] = {0, 1, 2, 3, 4, 5, 14};        // N0, NP, r6, r5, r4, r3 | cur: caNP
            FlatState2<FD> f;
            flat2_init<FD>(f, live, d, X, GE);
            flat2_loop<FD, CH>(1, nIter, nChunks, stage, f, X, GE,
                [&](int c, int i, auto setTag, const auto& mid) __attribute__((always_inline)) {
                    constexpr uint32_t SET = decltype(setTag)::value;      // the kinds this chunk loads (0: a steady chunk)
                    const ResPre<sig_t> q0 = res_pre<MODE, false>(f.ra[0], f.rb[0], f.rc[0], f.z1[0], f.z2[0]);      // N0's a comes from the track
                    const ResPre<sig_t> q1 = res_pre<MODE, ((SET >> 1) & 1u) != 0u>(f.ra[1], f.rb[1], f.rc[1], f.z1[1], f.z2[1]);
                    const ResPre<sig_t> q2 = res_pre<MODE, ((SET >> 2) & 1u) != 0u>(f.ra[2], f.rb[2], f.rc[2], f.z1[2], f.z2[2]);
                    const ResPre<sig_t> q3 = res_pre<MODE, ((SET >> 3) & 1u) != 0u>(f.ra[3], f.rb[3], f.rc[3], f.z1[3], f.z2[3]);
                    const ResPre<sig_t> q4 = res_pre<MODE, ((SET >> 4) & 1u) != 0u>(f.ra[4], f.rb[4], f.rc[4], f.z1[4], f.z2[4]);
                    const ResPre<sig_t> q5 = res_pre<MODE, ((SET >> 5) & 1u) != 0u>(f.ra[5], f.rb[5], f.rc[5], f.z1[5], f.z2[5]);
                    const sig_t caNP = f.cur[0];
                    mid(q0, q1, q2, q3, q4, q5, caNP);
                    const sig_t x = PIPE(pipeX, c, i);
                    sig_t zin = f.z1[0];
                    const sig_t n0 = res_post<MODE>(q0, x, zin, f.z2[0]);       // the anti-resonator remembers its INPUT (:133)
                    f.z1[0] = x;
                    const sig_t np = res_post<MODE>(q1, n0, f.z1[1], f.z2[1]);
                    sig_t o = fade_value(x, np, caNP);
                    o = res_post<MODE>(q2, o, f.z1[2], f.z2[2]);
                    o = res_post<MODE>(q3, o, f.z1[3], f.z2[3]);
                    o = res_post<MODE>(q4, o, f.z1[4], f.z2[4]);
                    o = res_post<MODE>(q5, o, f.z1[5], f.z2[5]);
                    PIPE(pipeO, c, i) = o;
                },
                noChunk);
#endif
        }
    } else if (FLAT && stage == 3) {
        // ================= flat S3: frication noise, parallel r1..r4 partial sum =================
        if constexpr (FLAT) {
            using FD = FlatDesc<3, 4, 3, false, 0x77u>;                // usually parallel 1..3 and the gains
            constexpr int GE[7] = {8, 9, 10, 11, 17, 18, 19};     // parallel 1..4 | cur: fricationAmplitude, preFormantGain, pa1..pa4
            FlatState2<FD> f;
            flat2_init<FD>(f, live, d, X, GE);
            sig_t fricNoise = 0;
            uint32_t noiseSt = noise_step(noise_first(nkey, ninc), ninc);     // frication: values 1, 3, 5, ...
            flat2_loop<FD, CH>(1, nIter, nChunks, stage, f, X, GE,
                [&](int c, int i, auto setTag, const auto& mid) __attribute__((always_inline)) {
                    constexpr uint32_t SET = decltype(setTag)::value;      // the kinds this chunk loads (0: a steady chunk)
                    const ResPre<sig_t> q0 = res_pre<MODE, ((SET >> 0) & 1u) != 0u>(f.ra[0], f.rb[0], f.rc[0], f.z1[0], f.z2[0]);
                    const ResPre<sig_t> q1 = res_pre<MODE, ((SET >> 1) & 1u) != 0u>(f.ra[1], f.rb[1], f.rc[1], f.z1[1], f.z2[1]);
                    const ResPre<sig_t> q2 = res_pre<MODE, ((SET >> 2) & 1u) != 0u>(f.ra[2], f.rb[2], f.rc[2], f.z1[2], f.z2[2]);
                    const ResPre<sig_t> q3 = res_pre<MODE, ((SET >> 3) & 1u) != 0u>(f.ra[3], f.rb[3], f.rc[3], f.z1[3], f.z2[3]);
                    fricNoise = (sig_t)noise_uniform(noiseSt) + (sig_t)0.75 * fricNoise;
                    noiseSt = noise_step2(noiseSt, ninc2);
                    const sig_t fric = fricNoise * (sig_t)0.3 * f.cur[0];
                    const sig_t y = (fric * f.cur[1]) * (sig_t)0.5;
                    const sig_t pa1 = f.cur[2], pa2 = f.cur[3], pa3 = f.cur[4], pa4 = f.cur[5];
                    mid(q0, q1, q2, q3, y, pa1, pa2, pa3, pa4);
                    sig_t par = 0;
                    sig_t w = res_post<MODE>(q0, y, f.z1[0], f.z2[0]);
                    par += (w - y) * pa1;
                    w = res_post<MODE>(q1, y, f.z1[1], f.z2[1]);
                    par += (w - y) * pa2;
                    w = res_post<MODE>(q2, y, f.z1[2], f.z2[2]);
                    par += (w - y) * pa3;
                    w = res_post<MODE>(q3, y, f.z1[3], f.z2[3]);
                    par += (w - y) * pa4;
                    PIPE(pipeA, c, i) = y; PIPE(pipeB, c, i) = par;
                },
                noChunk, make_sync(std::integral_constant<int, L::kBufs>{}, I0{}, I1{}, 0, 0, 4));
        }
    } else if (FLAT && (KLATT_FLAT_EXHAUSTIVE || stage == 2)) {
        // ================= flat final stage: r3, r2, r1 | parallel 5, 6, bypass | gain, clip, int16 -> PCM =================
        if constexpr (FLAT) {
#if KLATT_FLAT_LAYOUT == 2
            constexpr int GE[4] = {12, 13, 15, 16};               // parallel 5, 6 | cur: pa5, pa6, parallelBypass, outputGain
            using FD = FlatDesc<2, 2, 2, false, 0xEu>;                 // usually parallel 6 and the gains
            constexpr int RS = 0;                                      // cascade resonators in this stage
#else
            constexpr int GE[6] = {6, 7, 12, 13, 15, 16};         // r2, r1, parallel 5, 6 | cur: pa5, pa6, parallelBypass, outputGain
            using FD = FlatDesc<2, 4, 2, false, 0x3Bu>;                // usually c2, c1, parallel 6 and the gains
            constexpr int RS = 2;
#endif
            FlatState2<FD> f;
            flat2_init<FD>(f, live, d, X, GE);
            int16_t* const myRow = reinterpret_cast<int16_t*>(tile + lane * kTileStride);
            uint32_t it = 0;
            auto flush_tile = [&](uint32_t tileStart, uint32_t validTo) __attribute__((always_inline)) {
                rowCount[lane] = f.produced;
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // this wave alone owns the tile: wave-level ordering is enough
                constexpr int kChunksPerRow = kTile / 8;
                constexpr int kRowsPerPass = kLanes / kChunksPerRow;
                constexpr int kPasses = kLanes / kRowsPerPass;
                const int chunk = lane % kChunksPerRow;
                const uint32_t first = tileStart + (uint32_t)chunk * 8u;
                uint2 lo[kPasses], hi[kPasses];
                uint32_t cnt[kPasses];
                long long base[kPasses];
#pragma unroll
                for (int p = 0; p < kPasses; ++p) {
                    const int row = p * kRowsPerPass + lane / kChunksPerRow;
                    const uint2* src = reinterpret_cast<const uint2*>(tile + row * kTileStride + chunk * 16);
                    lo[p] = src[0]; hi[p] = src[1];
                    cnt[p] = rowCount[row];
                    base[p] = rowBase[row];
                }
#pragma unroll
                for (int p = 0; p < kPasses; ++p) {
                    if (cnt[p] > first && first < validTo) {
                        uint4* dst = reinterpret_cast<uint4*>(A.pcm + base[p] + first);
                        *dst = make_uint4(lo[p].x, lo[p].y, hi[p].x, hi[p].y);
                    }
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            };
            flat2_loop<FD, CH>(2, nIter, nChunks, stage, f, X, GE,
                [&](int c, int i, auto setTag, const auto& mid) __attribute__((always_inline)) {
                    constexpr uint32_t SET = decltype(setTag)::value;      // the kinds this chunk loads (0: a steady chunk)
                    // resonators of the stage: RS of the cascade's last ones, then parallel 5 and 6
                    constexpr int NR = 2 + RS;
                    ResPre<sig_t> q[4];
#pragma unroll
                    for (int r = 0; r < NR; ++r) q[r] = ((SET >> r) & 1u) ? res_pre<MODE, true>(f.ra[r], f.rb[r], f.rc[r], f.z1[r], f.z2[r])
                                                                          : res_pre<MODE, false>(f.ra[r], f.rb[r], f.rc[r], f.z1[r], f.z2[r]);
                    const sig_t pa5 = f.cur[0], pa6 = f.cur[1], bypass = f.cur[2], outGain = f.cur[3];
                    if constexpr (RS == 2) mid(q[0], q[1], q[2], q[3], pa5, pa6, bypass, outGain);
                    else mid(q[0], q[1], pa5, pa6, bypass, outGain);
                    sig_t o = PIPE(pipeO, c, i);
                    const sig_t y = PIPE(pipeA, c, i);
#pragma unroll
                    for (int r = 0; r < RS; ++r) o = res_post<MODE>(q[r], o, f.z1[r], f.z2[r]);
                    sig_t par = PIPE(pipeB, c, i);
                    sig_t w = res_post<MODE>(q[RS], y, f.z1[RS], f.z2[RS]);
                    par += (w - y) * pa5;
                    w = res_post<MODE>(q[RS + 1], y, f.z1[RS + 1], f.z2[RS + 1]);
                    par += (w - y) * pa6;
                    par = fade_value(par, y, bypass);
                    const sig_t mix = o + par;
                    const sig_t v = (mix * outGain) * (sig_t)4000.0;
                    const sig_t lo = (v < (sig_t)32000.0) ? v : (sig_t)32000.0;       // windows.h min(): NaN -> 32000
                    const sig_t cl = (lo > (sig_t)-32000.0) ? lo : (sig_t)-32000.0;
                    myRow[(it % kTile) + i] = (int16_t)(uint32_t)(int)cl;   // (int) truncates toward zero (:208)
                },
                [&]() __attribute__((always_inline)) { it += kChunk; if ((it % kTile) == 0) flush_tile(it - kTile, it); },
                make_sync(std::integral_constant<int, L::kBufs>{}, I2{}, I0{}, 2, 4, 0));
            if ((it % kTile) != 0) flush_tile(it - (it % kTile), it);
        }
    } else if (!NOISE && !NASAL && (stage == 1 || stage == 2)) {
        // ================= quiet, nasal-free S1: r6, r5, r4 and S2: r3, r2, r1 =================
        using D = StageDesc<6, 3, -1, false, false>;
        const bool s1 = (stage == 1);
        const int P[6] = {s1 ? 12 : 9, s1 ? 20 : 17, s1 ? 11 : 8, s1 ? 19 : 16, s1 ? 10 : 7, s1 ? 18 : 15};
        constexpr int RF[3] = {0, 2, 4}, RB[3] = {1, 3, 5};
        StageFrame<6, 3> f;
        stage_frame_init(f, live, lds + (s1 ? L::kFrames1 : L::kFrames2), lane);
        PipeT* const pin = s1 ? pipeX : pipeO;
        PipeT* const pout = s1 ? pipeO : pipeA;
        constexpr int nbuf_pin = 2, nbuf_pout = 2;      // (quiet launches: two buffers per pipe)
        auto dsp = [&](double o) __attribute__((always_inline)) -> double {
#pragma unroll
            for (int r = 0; r < 3; ++r) o = resonate<MODE>(f.z1[r], f.z2[r], f.ra[r], f.rb[r], f.rc[r], o);
            return o;
        };
        auto run = [&](int depth) __attribute__((always_inline)) {
            stage_loop<D, MODE, CH, KFil>(depth, nIter, nChunks, fullChunks, stage, f, nullptr, nullptr, noDelay, P, RF, RB, X, never, noBegin, noAlt, noAltSample,
                [&](int c, int i) __attribute__((always_inline)) { return PIPE(pin, c, i); },
                [&](int c, int i, bool steady, double pre) __attribute__((always_inline)) { PIPE(pout, c, i) = dsp((kPre && steady) ? pre : PIPE(pin, c, i)); },
                noFadeAlt,
                [&](int c, int i, bool emit) __attribute__((always_inline)) { if (emit) PIPE(pout, c, i) = dsp(PIPE(pin, c, i)); },
                nothing, nothing, noChunk);
        };
        if (s1) run(1); else run(2);
    } else if (stage == 1) {
        // ================= S1: N0 (anti), NP mixed by caNP, r6 [, r5, r4] =================
        constexpr bool HEAD = (SPLIT >> 2) == 0;           // N0 and NP are here (else in the source stage, and this stage is r6 ..)
        constexpr int NCASC = NOISE ? 3 + MOVED : 1;       // cascade resonators here: r6 [, r5, r4 [, r3 [, r2 [, r1]]]]
        constexpr int NR = (HEAD ? 2 : 0) + NCASC;
        using D = StageDesc<HEAD ? 2 * NR + 1 : 2 * NR, NR, -1, false, HEAD, NOISE>;
        // parameter list: (f, bw) of [N0, NP,] r6 .., [then caNP]
        constexpr int P[15] = {head_param(HEAD, NCASC, 0), head_param(HEAD, NCASC, 1), head_param(HEAD, NCASC, 2), head_param(HEAD, NCASC, 3), head_param(HEAD, NCASC, 4),
                               head_param(HEAD, NCASC, 5), head_param(HEAD, NCASC, 6), head_param(HEAD, NCASC, 7), head_param(HEAD, NCASC, 8), head_param(HEAD, NCASC, 9),
                               head_param(HEAD, NCASC, 10), head_param(HEAD, NCASC, 11), head_param(HEAD, NCASC, 12), head_param(HEAD, NCASC, 13), head_param(HEAD, NCASC, 14)};
        constexpr int RF[7] = {0, 2, 4, 6, 8, 10, 12};
        constexpr int RB[7] = {1, 3, 5, 7, 9, 11, 13};
        constexpr int CANP = HEAD ? 2 * NR : 0;
        StageFrame<D::NPARAM, NR> f;
        stage_frame_init(f, live, lds + L::kFrames1, lane);
        constexpr int GR1[7] = {HEAD ? 0 : 2, HEAD ? 1 : 3, HEAD ? 2 : 4, HEAD ? 3 : 5, HEAD ? 4 : 6, HEAD ? 5 : 7, 6};
        if (STREAM && live) stage_state_load<D>(f, nullptr, streamState, P, RF, RB, GR1, streamPurge);
        auto dsp = [&](double x) __attribute__((always_inline)) -> double {
            double o = x;
            if (HEAD) {
                const double n0 = dot3<MODE>(f.ra[0], x, f.rb[0], f.z1[0], f.rc[0], f.z2[0]);
                f.z2[0] = f.z1[0]; f.z1[0] = x;                       // anti-resonator remembers its INPUT (:133)
                const double np = resonate<MODE>(f.z1[1], f.z2[1], f.ra[1], f.rb[1], f.rc[1], n0);
                o = fade_value(x, np, f.cur[CANP]);
            }
#pragma unroll
            for (int r = HEAD ? 2 : 0; r < NR; ++r) o = resonate<MODE>(f.z1[r], f.z2[r], f.ra[r], f.rb[r], f.rc[r], o);
            return o;
        };
        stage_loop<D, MODE, CH, KFil>(1, nIter, nChunks, fullChunks, stage, f, nullptr, nullptr, noDelay, P, RF, RB, X, never, noBegin, noAlt, noAltSample,
            [&](int c, int i) __attribute__((always_inline)) { return PIPE(pipeX, c, i); },
            [&](int c, int i, bool steady, double pre) __attribute__((always_inline)) { PIPE(pipeO, c, i) = dsp((kPre && steady) ? pre : PIPE(pipeX, c, i)); },
            noFadeAlt,
            [&](int c, int i, bool emit) __attribute__((always_inline)) { if (emit) PIPE(pipeO, c, i) = dsp(PIPE(pipeX, c, i)); },
            nothing, nothing, noChunk);
        if (STREAM && live) stage_state_save<D>(f, nullptr, streamState, P, GR1);
    } else if (NOISE && stage == 3) {
        // ================= noisy S3: frication noise, parallel r1..r4 partial sum =================
        // tracked: (pf, pb) of parallel 1..4, then 24 fricationAmplitude, 44 preFormantGain, pa1..4 (37..40)
        using D = StageDesc<14, 4, 9, false, false, true>;
        constexpr int P[14] = {25, 31, 26, 32, 27, 33, 28, 34, 24, 44, 37, 38, 39, 40};
        constexpr int RF[4] = {0, 2, 4, 6}, RB[4] = {1, 3, 5, 7};
        StageFrame<14, 4> f;
        stage_frame_init(f, live, lds + L::kFrames3, lane);
        double fricNoise = 0.0;
        uint32_t noiseSt = noise_step(noise_first(nkey, ninc), ninc);     // frication: values 1, 3, 5, ...
        constexpr int GR3[4] = {8, 9, 10, 11};
        if (STREAM && live) {
            if (streamState[239] != 0.0) { fricNoise = streamState[211]; noiseSt = noise_step((uint32_t)streamState[221], ninc); }   // slot 221: the state of the next aspiration value
            stage_state_load<D>(f, nullptr, streamState, P, RF, RB, GR3, streamPurge);
        }
        auto dsp = [&](int c, int i) __attribute__((always_inline)) {
            fricNoise = noise_uniform(noiseSt) + 0.75 * fricNoise;
            noiseSt = noise_step2(noiseSt, ninc2);
            const double fric = fricNoise * 0.3 * f.cur[8];
            const double y = (fric * f.cur[9]) * 0.5;
            double par = 0.0;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const double w = resonate<MODE>(f.z1[r], f.z2[r], f.ra[r], f.rb[r], f.rc[r], y);
                par += (w - y) * f.cur[10 + r];
            }
            PIPE(pipeA, c, i) = y; PIPE(pipeB, c, i) = par;
        };
        stage_loop<D, MODE, CH, KSrc>(1, nIter, nChunks, fullChunks, stage, f, nullptr, nullptr, noDelay, P, RF, RB, X, never, noBegin, noAlt, noAltSample,
            [&](int, int) __attribute__((always_inline)) { return 0.0; },
            [&](int c, int i, bool, double) __attribute__((always_inline)) { dsp(c, i); },
            noFadeAlt,
            [&](int c, int i, bool emit) __attribute__((always_inline)) { if (emit) dsp(c, i); },
            nothing, nothing, noChunk);
        if (STREAM && live) { stage_state_save<D>(f, nullptr, streamState, P, GR3); streamState[211] = fricNoise; }
    } else if (!NOISE && stage == 2) {
        // ================= quiet S2: r5, r4, r3 =================
        using D = StageDesc<6, 3, -1, false, false>;
        constexpr int P[6] = {11, 19, 10, 18, 9, 17};
        constexpr int RF[3] = {0, 2, 4}, RB[3] = {1, 3, 5};
        StageFrame<6, 3> f;
        stage_frame_init(f, live, lds + L::kFrames2, lane);
        auto dsp = [&](double o) __attribute__((always_inline)) -> double {
#pragma unroll
            for (int r = 0; r < 3; ++r) o = resonate<MODE>(f.z1[r], f.z2[r], f.ra[r], f.rb[r], f.rc[r], o);
            return o;
        };
        stage_loop<D, MODE, CH, KFil>(2, nIter, nChunks, fullChunks, stage, f, nullptr, nullptr, noDelay, P, RF, RB, X, never, noBegin, noAlt, noAltSample,
            [&](int c, int i) __attribute__((always_inline)) { return PIPE(pipeO, c, i); },
            [&](int c, int i, bool steady, double pre) __attribute__((always_inline)) { PIPE(pipeA, c, i) = dsp((kPre && steady) ? pre : PIPE(pipeO, c, i)); },
            noFadeAlt,
            [&](int c, int i, bool emit) __attribute__((always_inline)) { if (emit) PIPE(pipeA, c, i) = dsp(PIPE(pipeO, c, i)); },
            nothing, nothing, noChunk);
    } else {
        // ================= final stage: rest of the cascade, (parallel r5, r6 + bypass), gain, clip, PCM ===
        // noisy (stage 2): r3, r2, r1 | parallel 5, 6 | pa5, pa6, parallelBypass, outputGain
        // quiet (stage 3): r2, r1 | outputGain          quiet, nasal-free (stage 3): outputGain only
        constexpr int NC = NOISE ? 3 - MOVED : (NASAL ? 2 : 0);   // cascade resonators here
        constexpr int NR = NOISE ? NC + 2 : NC;
        constexpr int NPAR = NOISE ? 2 * NC + 8 : (NASAL ? 5 : 1);
        using D = StageDesc<NPAR, NR, -1, false, false, NOISE>;
        // noisy: (f, bw) of the cascade resonators left here, of parallel 5, 6, then pa5, pa6, parallelBypass, outputGain
        constexpr int P[14] = {NOISE ? final_param(MOVED, 0) : (NASAL ? 8 : 45), NOISE ? final_param(MOVED, 1) : 16, NOISE ? final_param(MOVED, 2) : 7,
                               NOISE ? final_param(MOVED, 3) : 15, NOISE ? final_param(MOVED, 4) : 45, final_param(MOVED, 5), final_param(MOVED, 6), final_param(MOVED, 7),
                               final_param(MOVED, 8), final_param(MOVED, 9), final_param(MOVED, 10), final_param(MOVED, 11), final_param(MOVED, 12), 45};
        constexpr int RF[5] = {0, 2, NOISE ? 4 : 0, 6, 8};
        constexpr int RB[5] = {1, 3, NOISE ? 5 : 0, 7, 9};
        constexpr int PA = 2 * NC + 4;                    // noisy: pa5, pa6, parallelBypass follow the resonators' parameters
        constexpr int OUTGAIN = NOISE ? 2 * NC + 7 : (NASAL ? 4 : 0);
        StageFrame<NPAR, NR> f;
        stage_frame_init(f, live, lds + (NOISE ? L::kFrames2 : L::kFrames3), lane);
        constexpr int GRF[5] = {final_res(MOVED, 0), final_res(MOVED, 1), final_res(MOVED, 2), final_res(MOVED, 3), 13};
        if (STREAM && live) stage_state_load<D>(f, nullptr, streamState, P, RF, RB, GRF, streamPurge);
        int16_t* const myRow = reinterpret_cast<int16_t*>(tile + lane * kTileStride);

        auto finish = [&](double o, double y, double part) __attribute__((always_inline)) -> uint32_t {
#pragma unroll
            for (int r = 0; r < NC; ++r) o = resonate<MODE>(f.z1[r], f.z2[r], f.ra[r], f.rb[r], f.rc[r], o);
            double mix = o;
            if (NOISE) {
                double par = part;
#pragma unroll
                for (int r = NC; r < NC + 2; ++r) {
                    const double w = resonate<MODE>(f.z1[r], f.z2[r], f.ra[r], f.rb[r], f.rc[r], y);
                    par += (w - y) * f.cur[PA + (r - NC)];
                }
                par = fade_value(par, y, f.cur[PA + 2]);
                mix = o + par;
            }
            const double v = (mix * f.cur[OUTGAIN]) * 4000.0;
            const double lo = (v < 32000.0) ? v : 32000.0;       // windows.h min(): NaN -> 32000
            const double cl = (lo > -32000.0) ? lo : -32000.0;
            return (uint32_t)(int)cl;                             // (int) truncates toward zero (:208)
        };
        auto flush_tile = [&](uint32_t tileStart, uint32_t validTo) __attribute__((always_inline)) {
            rowCount[lane] = f.produced;
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // this wave alone owns the tile: wave-level ordering is enough
            constexpr int kChunksPerRow = kTile / 8;
            constexpr int kRowsPerPass = kLanes / kChunksPerRow;
            constexpr int kPasses = kLanes / kRowsPerPass;
            const int chunk = lane % kChunksPerRow;
            const uint32_t first = tileStart + (uint32_t)chunk * 8u;
            // every LDS read of all passes first (one wait), then the stores
            uint2 lo[kPasses], hi[kPasses];
            uint32_t cnt[kPasses];
            long long base[kPasses];
#pragma unroll
            for (int p = 0; p < kPasses; ++p) {
                const int row = p * kRowsPerPass + lane / kChunksPerRow;
                const uint2* src = reinterpret_cast<const uint2*>(tile + row * kTileStride + chunk * 16);
                lo[p] = src[0]; hi[p] = src[1];
                cnt[p] = rowCount[row];
                base[p] = rowBase[row];
            }
#pragma unroll
            for (int p = 0; p < kPasses; ++p) {
                if (cnt[p] > first && first < validTo) {
                    uint4* dst = reinterpret_cast<uint4*>(A.pcm + base[p] + first);
                    *dst = make_uint4(lo[p].x, lo[p].y, hi[p].x, hi[p].y);
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        };

        uint32_t it = 0;   // samples stepped so far (wave-uniform); the chunk being processed starts at `it`
        auto in0 = [&](int c, int i) __attribute__((always_inline)) -> double { return NOISE ? PIPE(pipeO, c, i) : PIPE(pipeA, c, i); };
        auto inY = [&](int c, int i) __attribute__((always_inline)) -> double { return NOISE ? PIPE(pipeA, c, i) : 0.0; };
        auto inP = [&](int c, int i) __attribute__((always_inline)) -> double { return NOISE ? PIPE(pipeB, c, i) : 0.0; };
        stage_loop<D, MODE, CH, KFil>((NOISE ? 2 : 3), nIter, nChunks, fullChunks, stage, f, nullptr, nullptr, noDelay, P, RF, RB, X, never, noBegin, noAlt, noAltSample,
            in0,
            [&](int c, int i, bool steady, double pre) __attribute__((always_inline)) {
                myRow[(it % kTile) + i] = (int16_t)finish((kPre && steady) ? pre : in0(c, i), inY(c, i), inP(c, i));
            },
            noFadeAlt,
            [&](int c, int i, bool emit) __attribute__((always_inline)) {
                if (emit) { myRow[(it % kTile) + i] = (int16_t)finish(in0(c, i), inY(c, i), inP(c, i)); f.produced++; }
            },
            [&](int n) __attribute__((always_inline)) { f.produced += n; },
            [&](int n) __attribute__((always_inline)) { f.produced += n; },
            [&]() __attribute__((always_inline)) { it += kChunk; if ((it % kTile) == 0) flush_tile(it - kTile, it); });
        if ((it % kTile) != 0) flush_tile(it - (it % kTile), it);
        if (STREAM && live) stage_state_save<D>(f, nullptr, streamState, P, GRF);
    }
#undef PIPE
}

}  // namespace klatt
