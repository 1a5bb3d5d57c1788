// klatt_lanepipe.h -- lane-pipelined batch kernel for quiet, nasal-free utterances.
//
// The stage-parallel kernel (klatt_systolic.h) cuts the per-sample chain of the reference's generate() loop
// (reference src/speechWaveGenerator.cpp:197-214) into four wavefronts over 64 utterances; the time a launch
// takes is (samples per utterance) x (time per sample of the slowest stage), however few utterances there are,
// so a 4096-utterance batch keeps 64 of 256 CUs busy at ~65 ns per sample.  Here the cascade r6 -> r5 -> ... -> r1
// (reference :149-156) is laid across the LANES of a wavefront instead: lane (g, k) runs resonator k of
// utterance g, and hands its output to lane (g, k + 1) with one DPP rotate per step (no LDS, no barrier):
//
//   wave 0      S0   frame(0, 1-6, 44) + glottal source for the workgroup's 16 utterances      -> x   (LDS pipe)
//   wave 1, 2   F    8 utterances each, two to a 16-lane row: row position 2 k + j = resonator r(6-k) of the row's utterance j
//                    step t: resonator k filters sample t - k; in = (k == 0) ? x[t] : out of resonator k - 1 (DPP row_shr:2)
//   wave 3      FIN  outputGain, x 4000, clip, int16 -> PCM tile -> HBM
//
// A step costs one resonator (5 f64 operations in the reference's order) plus the hand-over, ~3x less than a
// stage of the stage-parallel kernel, and a 4096-utterance batch becomes 256 workgroups.  Every lane runs its own
// copy of the frame state machine (reference src/frame.cpp:41-80) for its two parameters (f, bw), started k steps
// late; the arithmetic per sample is the lane kernel's, operation for operation, so the PCM is bit-identical.
//
// Eligible utterances (classified on the host, UTT_NO_NASAL): no noise (the quiet group) and caNP == 0 with
// bounded, stable N0/NP parameters in every frame.  Then the cascade input passes the nasal pair untouched
// (reference :151-152: lerp(x, NP(N0(x)), 0) == x + (np - x) * 0 == x for finite np), N0's and NP's memories are
// never observed, and both are skipped.
#pragma once

#include "klatt_systolic.h"

namespace klatt {

constexpr uint32_t UTT_NO_NASAL = 2u;     // UttDesc.flags: caNP == 0 throughout, N0/NP finite and stable

constexpr int kLpK = 6;                   // lanes (cascade resonators r6..r1) per utterance
constexpr int kLpUPR = 2;                 // utterances per 16-lane row, interleaved: row position p = 2 k + (utterance & 1); positions 12..15 idle
constexpr int kLpUPW = 4 * kLpUPR;        // utterances per filter wave (8)
constexpr int kLpUPG = 2 * kLpUPW;        // utterances per workgroup (two filter waves)
constexpr int kLpSkew = kLpK - 1;         // the last lane emits sample t - kLpSkew at step t

template <int CH>
struct LpLds {
    static constexpr int kPipeX = 0;                                   // S0 -> F : [2 buffers][CH][kLpUPG] f64
    static constexpr int kPipeY = kPipeX + 2 * CH * kLpUPG * 8;        // F -> FIN: ring [4][CH][kLpUPG] f64
    static constexpr int kTileOff = kPipeY + 4 * CH * kLpUPG * 8;
    static constexpr int kT = CH > kTile ? CH : kTile;                // samples per tile row: a whole chunk leaves at once
    static constexpr int kTStride = kT * 2 + 8;                        // bytes per row; the pad keeps ds_write_b16 conflict-free
    static constexpr int kRowBase = kTileOff + kLanes * kTStride;
    static constexpr int kRowCount = kRowBase + kLanes * 8;
    static constexpr int kMaxLen = kRowCount + kLanes * 4;
    static constexpr int kFrames0 = kMaxLen + 16;                      // S0: old values of 7 parameters (targets in registers)
    static constexpr int kFramesF = kFrames0 + 7 * kLanes * 8;         // per filter wave: old + new of (f, bw)
    static constexpr int kFramesFin = kFramesF + 2 * (2 * 2 * kLanes * 8);
    static constexpr int kBytes = kFramesFin + 2 * 1 * kLanes * 8;     // FIN: old + new of outputGain
};

// The hand-over of a step: row position p receives `out` of position p - 2 (the same utterance's previous resonator); positions 0 and 1,
// which have no such neighbour, keep `x`, their utterance's cascade input -- v_mov_b32_dpp row_shr:2 with bound_ctrl off, twice for a
// double, and no select (lanes laid out six to an utterance needed wave_ror:1 plus two v_cndmask for the first lanes: 11 VALU
// instructions per step instead of 9).
__device__ __forceinline__ double lp_hand_over(double out, double x)
{
    int lo = __builtin_amdgcn_update_dpp(__double2loint(x), __double2loint(out), 0x112, 0xF, 0xF, false);
    int hi = __builtin_amdgcn_update_dpp(__double2hiint(x), __double2hiint(out), 0x112, 0xF, 0xF, false);
    return __hiloint2double(hi, lo);
}

// The chunk loop of every pipeline element is stage_loop of klatt_systolic.h (steady / fading / sample-by-sample paths, one barrier
// per chunk).  The steady paths here are branch-free so that a whole chunk is one basic block (no EXEC changes: a masked store
// costs a lone wave ~11 ns); tools/ubench_lanepipe.hip, tools/len_probe.py and DESIGN.md section 4.3 have the measurements.
template <int MODE, int CH, int WPS>
__global__ void __launch_bounds__(kLanes * kStages, WPS) klatt_lanepipe(const KernelArgs A)
{
    using L = LpLds<CH>;
    static_assert((CH & (CH - 1)) == 0 && CH > kLpSkew, "CH must be a power of two above the skew");
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    double* const pipeX = reinterpret_cast<double*>(lds + L::kPipeX);
    double* const pipeY = reinterpret_cast<double*>(lds + L::kPipeY);
    unsigned char* const tile = lds + L::kTileOff;
    long long* const rowBase = reinterpret_cast<long long*>(lds + L::kRowBase);
    uint32_t* const rowCount = reinterpret_cast<uint32_t*>(lds + L::kRowCount);
    uint32_t* const maxLenP = reinterpret_cast<uint32_t*>(lds + L::kMaxLen);

    const int lane = threadIdx.x & (kLanes - 1);
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const bool filter = (wave == 1 || wave == 2);
    // which utterance of the workgroup this lane works for, and (filter waves) which resonator
    const int rowPos = lane & 15;
    const int k = filter ? (rowPos >> 1) : 0;
    const int uw = filter ? (wave - 1) * kLpUPW + (lane >> 4) * kLpUPR + (rowPos & 1) : lane;
    const bool seated = filter ? (k < kLpK) : (lane < kLpUPG);
    const long long slot = (long long)blockIdx.x * kLpUPG + uw;
    const uint32_t u = (seated && slot < A.nSlots) ? A.order[slot] : 0xFFFFFFFFu;
    const bool live = (u != 0xFFFFFFFFu);

    UttDesc d;
    d.frameStart = 0; d.outStart = 0; d.nFrames = 0; d.seed = 0; d.flags = 0; d.length = 0;
    if (live) d = A.utt[u];
    const StageCtx X{A, d, A.frames + d.frameStart * kNumParams, A.meta + d.frameStart, 0u, 0xFFFFFFFFu};
    const double srF = A.sampleRateF, invSr = A.invSampleRate;   // by value into the lambdas below
    int16_t* const pcmOut = A.pcm;

    if (threadIdx.x == 0) *maxLenP = 0;
    __syncthreads();
    if (wave == 0) atomicMax(maxLenP, d.length);
    __syncthreads();
    const uint32_t maxLen = *maxLenP;
    const int nChunks = (int)((maxLen + (uint32_t)kLpSkew + CH - 1) / CH);   // steps = samples + skew
    const int nIter = nChunks + 3;                                           // S0 depth 0, F depth 1, FIN depth 3
    const int uwSafe = seated ? uw : 0;

    // S0 writes x of sample (c, i) into buffer c & 1; F reads it one iteration later
#define LP_X(c, i) pipeX[(((c) & 1) * CH + (i)) * kLpUPG + uwSafe]
    // F's last lane writes at step t = c * CH + i the cascade output of sample t - kLpSkew into ring slot t; FIN reads
    // sample s from slot s + kLpSkew, two iterations after F wrote the later of the two chunks that can hold it
#define LP_Y(t) pipeY[((t) & (4 * CH - 1)) * kLpUPG + uwSafe]

    if (wave == 0) {
        // ================= S0: frame + glottal source (quiet: no aspiration, no turbulence) =================
        using D = StageDesc<7, 0, 6, true, false>;
        constexpr int P[7] = {1, 2, 3, 4, 5, 6, 44};
        constexpr int RF[1] = {0}, RB[1] = {0};
        StageFrame<7, 0, true> f;
        PitchState ps;
        stage_frame_init(f, live, lds + L::kFrames0, lane);
        ps.cur0 = 0.0; ps.old0 = 0.0; ps.new0 = 0.0; ps.oldInc = 0.0; ps.newInc = 0.0;
        double pitchPhase = 0.0, vibPhase = 0.0;
        int32_t lastIndex = -1;
        uint32_t delay = 0;
        bool vibFrames = false;
        double incConst = 0.0;       // the phase increment of a steady chunk in which no live lane's pitch glides

        auto finishSource = [&]() __attribute__((always_inline)) -> double {
            const double voice = (pitchPhase * 2.0) - 1.0;
            const double src = voice * f.cur[4];           // all noise gains are zero: turbulence and aspiration add exactly +0
            return (src * f.cur[6]) * 0.5;
        };
        auto source = [&](bool waveVib) __attribute__((always_inline)) -> double {
            double vib = 1.0;
            if (waveVib) {
                const double vs = f.cur[1];
                const double adv = frac_toward_zero(div_by(vs, srF, invSr) + vibPhase);
                vibPhase = (vs != 0.0) ? adv : vibPhase;
                vib = (sin(vibPhase * 6.283185307179586) * 0.06 * f.cur[0]) + 1.0;
            }
            pitchPhase = frac_toward_zero(div_by(ps.cur0 * vib, srF, invSr) + pitchPhase);
            return finishSource();
        };
        auto vib_live_now = [&]() __attribute__((always_inline)) -> bool {
            return vibFrames || f.cur[0] != 0.0 || f.cur[1] != 0.0 || vibPhase != vibPhase;
        };
        using K0 = LoopKnobs<false, true, true, false, false, CH>;
        stage_loop<D, MODE, CH, K0>(0, nIter, nChunks, nChunks, wave, f, &ps, &lastIndex, delay, P, RF, RB, X,
            [&]() { return __any(!f.done && vib_live_now()); },
            [&](int kind) -> bool {
                if (kind != 0 || __any(!f.done && ps.oldInc != 0.0)) return false;
                // no live lane's pitch glides in this steady chunk: cur0 + 0 repeated is cur0 + 0 once, and the
                // phase increment (cur0 * 1) / sr is the same for every sample
                ps.cur0 += ps.oldInc;
                incConst = div_by(ps.cur0 * 1.0, srF, invSr);
                return true;
            },
            [&](int c) {
#pragma unroll
                for (int i = 0; i < CH; ++i) {
                    pitchPhase = frac_toward_zero(incConst + pitchPhase);
                    LP_X(c, i) = finishSource();
                }
            },
            [&](int, int, double) {},
            [&](int, int) { return 0.0; },
            [&](int c, int i, bool steady, double) {
                if (steady) ps.cur0 += ps.oldInc;
                LP_X(c, i) = source(false);
            },
            // a fading chunk in which only the gain (and the pitch) move and no target is NaN: fades into and out of
            // silence (reference src/frame.cpp:59-67); same operations as stage_fade + source with the differences and
            // the LDS values taken once (see s0_fade_alt in klatt_systolic.h)
            [&](int c, bool lerp, bool gainOnly) -> bool {
                if (!gainOnly) return false;
                constexpr int GI = 6;
                const double g0 = lerp ? f.oldL[GI * kLanes] : f.cur[GI], g1 = lerp ? f.getNew(GI) : f.cur[GI];
                if (__any(g1 != g1 || ps.new0 != ps.new0)) return false;
                const double gd = g1 - g0, p0 = ps.old0, pd = ps.new0 - p0, nf = (double)f.newFade;
                if (!__any(pd != 0.0 || p0 != p0)) {
                    ps.cur0 = p0 + 0.0;
                    const double inc = div_by(ps.cur0 * 1.0, srF, invSr);
#pragma unroll
                    for (int i = 0; i < CH; ++i) {
                        f.cnt++;
                        const double ratio = div_by((double)f.cnt, nf, f.invFade);
                        const double gain = g0 + (gd * ratio);
                        pitchPhase = frac_toward_zero(inc + pitchPhase);
                        LP_X(c, i) = ((((pitchPhase * 2.0) - 1.0) * f.cur[4]) * gain) * 0.5;
                        if (i == CH - 1) f.cur[GI] = gain;
                    }
                } else {
#pragma unroll
                    for (int i = 0; i < CH; ++i) {
                        f.cnt++;
                        const double ratio = div_by((double)f.cnt, nf, f.invFade);
                        ps.cur0 = p0 + (pd * ratio);
                        const double gain = g0 + (gd * ratio);
                        pitchPhase = frac_toward_zero(div_by(ps.cur0 * 1.0, srF, invSr) + pitchPhase);
                        LP_X(c, i) = ((((pitchPhase * 2.0) - 1.0) * f.cur[4]) * gain) * 0.5;
                        if (i == CH - 1) f.cur[GI] = gain;
                    }
                }
                return true;
            },
            [&](int c, int i, bool emit) {
                if (emit && f.hasNew && f.cnt == 0)
                    vibFrames = f.oldL[0] != 0.0 || f.oldL[kLanes] != 0.0 || f.getNew(0) != 0.0 || f.getNew(1) != 0.0;
                const bool waveVib = __any(emit && vib_live_now());
                if (emit) { LP_X(c, i) = source(waveVib); f.produced++; }
            },
            [&](int n) { ps.old0 = ps.cur0; f.produced += n; },
            [&](int n) { f.produced += n; },
            [&]() {});
        if (live) {
            UttResult res;
            res.produced = f.produced; res.framesTaken = f.nextFrame; res.lastIndex = lastIndex; res.drained = 1u;
            A.result[u] = res;
        }
    } else if (filter) {
        // ================= F: one cascade resonator per lane, r6 (k = 0) ... r1 (k = 5) =================
        using D = StageDesc<2, 1, -1, false, false>;
        const int P[2] = {12 - k, 20 - k};                   // cf6..cf1 = parameters 12..7, cb6..cb1 = 20..15
        constexpr int RF[1] = {0}, RB[1] = {1};
        StageFrame<2, 1> f;
        stage_frame_init(f, live, lds + L::kFramesF + (wave - 1) * (2 * 2 * kLanes * 8), lane);
        const bool last = (k == kLpK - 1);
        uint32_t delay = (uint32_t)k;
        double out = 0.0;            // this lane's latest output: what lane k + 1 reads next step
        // steady chunks keep the step outputs in registers and the last lanes store them all at the end of the chunk: one EXEC
        // change per chunk (an s_and_saveexec / s_or pair per step costs a lone wave ~11 ns), and only the 10 lanes that have
        // something to say write (every lane storing every step, 54 of them to a scratch area, was 17 % of a cfg1 launch:
        // 1 KB of LDS writes per step and filter wave, profiles/r3_cfg1_notes.txt)
        double keep[CH];
        using KF = LoopKnobs<true, true, true, false, false, CH>;
        stage_loop<D, MODE, CH, KF>(1, nIter, nChunks, nChunks, wave, f, nullptr, nullptr, delay, P, RF, RB, X,
            [&]() { return __any(!f.done && delay > 0u); },
            [&](int) -> bool { return true; },
            [&](int) {},
            [&](int c, int i, double pre) {
                const double in = lp_hand_over(out, pre);
                out = resonate<MODE>(f.z1[0], f.z2[0], f.ra[0], f.rb[0], f.rc[0], in);
                keep[i] = out;
                if (i == CH - 1 && last) {
#pragma unroll
                    for (int j = 0; j < CH; ++j) LP_Y(c * CH + j) = keep[j];
                }
            },
            [&](int c, int i) { return LP_X(c, i); },
            [&](int c, int i, bool, double) {
                const double in = lp_hand_over(out, LP_X(c, i));
                out = resonate<MODE>(f.z1[0], f.z2[0], f.ra[0], f.rb[0], f.rc[0], in);
                if (last) LP_Y(c * CH + i) = out;
            },
            [&](int, bool, bool) -> bool { return false; },
            [&](int c, int i, bool emit) {
                const double in = lp_hand_over(out, LP_X(c, i));            // every lane takes part in the shift
                if (emit) {
                    out = resonate<MODE>(f.z1[0], f.z2[0], f.ra[0], f.rb[0], f.rc[0], in);
                    if (last) LP_Y(c * CH + i) = out;
                }
            },
            [&](int) {}, [&](int) {}, [&]() {});
    } else {
        // ================= FIN: outputGain, x 4000, clip, int16, PCM tile (reference :207-208) =================
        using D = StageDesc<1, 0, -1, false, false>;
        constexpr int P[1] = {45};
        constexpr int RF[1] = {0}, RB[1] = {0};
        StageFrame<1, 0> f;
        stage_frame_init(f, live, lds + L::kFramesFin, lane);
        rowBase[lane] = d.outStart; rowCount[lane] = 0;     // read by this wave only
        int16_t* const myRow = reinterpret_cast<int16_t*>(tile + lane * L::kTStride);
        uint32_t delay = 0;
        uint32_t it = 0;             // samples stepped so far (wave-uniform)

        auto finish = [&](double o) __attribute__((always_inline)) -> uint32_t {
            const double v = (o * f.cur[0]) * 4000.0;
            const double lo = (v < 32000.0) ? v : 32000.0;       // windows.h min(): NaN -> 32000
            const double cl = (lo > -32000.0) ? lo : -32000.0;
            return (uint32_t)(int)cl;                             // (int) truncates toward zero (:208)
        };
        auto flush_tile = [&](uint32_t tileStart, uint32_t validTo) __attribute__((always_inline)) {
            rowCount[lane] = f.produced;
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // this wave alone owns the tile
            constexpr int kChunksPerRow = L::kT / 8;
            constexpr int kRowsPerPass = kLanes / kChunksPerRow;
            constexpr int kPasses = (kLpUPG + kRowsPerPass - 1) / kRowsPerPass;   // only the first kLpUPG rows are in use
            const int chunk = lane % kChunksPerRow;
            const uint32_t firstS = tileStart + (uint32_t)chunk * 8u;
            uint2 lo[kPasses], hi[kPasses];
            uint32_t cnt[kPasses];
            long long base[kPasses];
#pragma unroll
            for (int p = 0; p < kPasses; ++p) {
                const int row = p * kRowsPerPass + lane / kChunksPerRow;
                const uint2* src = reinterpret_cast<const uint2*>(tile + row * L::kTStride + chunk * 16);
                lo[p] = src[0]; hi[p] = src[1];
                cnt[p] = rowCount[row];
                base[p] = rowBase[row];
            }
#pragma unroll
            for (int p = 0; p < kPasses; ++p) {
                if (cnt[p] > firstS && firstS < validTo) {
                    uint4* dst = reinterpret_cast<uint4*>(pcmOut + base[p] + firstS);
                    *dst = make_uint4(lo[p].x, lo[p].y, hi[p].x, hi[p].y);
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        };
        using KE = LoopKnobs<true, true, true, false, false, CH>;
        stage_loop<D, MODE, CH, KE>(3, nIter, nChunks, nChunks, wave, f, nullptr, nullptr, delay, P, RF, RB, X,
            [&]() { return false; },
            [&](int) -> bool { return true; },
            [&](int) {},
            [&](int, int i, double pre) { myRow[(it % L::kT) + i] = (int16_t)finish(pre); },
            [&](int c, int i) { return LP_Y(c * CH + i + kLpSkew); },
            [&](int c, int i, bool, double) { myRow[(it % L::kT) + i] = (int16_t)finish(LP_Y(c * CH + i + kLpSkew)); },
            [&](int, bool, bool) -> bool { return false; },
            [&](int c, int i, bool emit) {
                if (emit) { myRow[(it % L::kT) + i] = (int16_t)finish(LP_Y(c * CH + i + kLpSkew)); f.produced++; }
            },
            [&](int n) { f.produced += n; },
            [&](int n) { f.produced += n; },
            [&]() { it += CH; if ((it % L::kT) == 0) flush_tile(it - L::kT, it); });
        if ((it % L::kT) != 0) flush_tile(it - (it % L::kT), it);
    }
#undef LP_X
#undef LP_Y
}

}  // namespace klatt
