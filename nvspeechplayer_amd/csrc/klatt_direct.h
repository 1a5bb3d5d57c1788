// klatt_direct.h -- direct stages: the stage-parallel kernel for batches whose fades share NOTHING and whose lanes fade at
// unrelated times (65 536 different sentences in different voices), and for batches that may not or cannot have tracks.
//
// The flat stages of klatt_systolic.h take every fade sample's coefficients from TRACKS: one dense evaluation per distinct fade of
// the batch, shared by every utterance that makes the transition.  That pays when fades are shared.  When they are not, the
// tracks are as large as the work itself (12 bytes per output sample), and before this file such batches fell back to stages that
// carry a replica of the frame state machine, two copies of the fade's end points in LDS and a wave-uniform decision tree per
// chunk (klatt_systolic.h, `stage_loop`): 60 ms for BASELINE configs[2]'s 65 536 utterances once their timings differ, against
// 8.5 ms for the benchmark's aligned copies.
//
// Here the stages keep the flat stages' CONTROL -- sample positions from the durations alone (frame k is dequeued on sample T_k,
// T_k+1 = T_k + max(min_k, fade_k + 1) + 1, its fade's samples are T_k + 1 .. T_k + fade_k; reference src/frame.cpp:41-80), one
// counter per lane, no dequeue / fade-end events, nothing in LDS but the pipes and the PCM tile -- and COMPUTE what the tracks
// would have held, behind a sample's filter arithmetic (DirectMid):
//
//   * per lane: the fade-sample index `cnt` of the values now in registers (cnt == F, the fade's length: the lane is not fading
//     and keeps the fade's last values, as the reference does until the next fade's first sample), and per parameter what the
//     running fade needs to advance it;
//   * ONE masked block per sample for the lanes with cnt < F (skipped when the wavefront has none): cnt + 1, and
//     MODE_EXACT: ratio = cnt / F (correctly rounded), value = from + ((to - from) * ratio) (reference src/frame.cpp:48-53,
//     src/utils.h:20-23), and for a resonator r = exp(-pi bw / sr), cs = cos(2 pi (-f) / sr), c = -(r r), b = r cs 2, a = 1 - b - c
//     (reference src/speechWaveGenerator.cpp:112-127) with the straight-line kernels of klatt_math.h (constants as scalar operands
//     of the FMAs), classified ONCE per chunk and wave (no range reduction / cosine quadrant -1 / general) from bits the seeds
//     carry.  Which kinds are evaluated is a wave-uniform mask per chunk (the OR over the lanes of what their running or starting
//     fades move): scalar branches, no ballots per sample.  The block is tied BEHIND the sample's arithmetic (an empty asm statement
//     makes the counter depend on the sample's last value): the compiler otherwise hoists it above the filters and keeps a
//     copy of every coefficient for the lanes it masks off;
//   * MODE_FAST replaces the polynomials by SURVEY section 7's recurrences: with f and bw linear in the fade-sample index,
//     the pole P = 2 r e^(i theta) advances by a constant complex factor w = q e^(i delta) per sample and r^2 by q^2:
//         P <- P w (4 operations), r^2 <- r^2 q^2, b = Re P, c = -r^2, a = 1 - b - c           (7 instead of ~36)
//     re-seeded EXACTLY at every fade's first sample (klatt_seeds evaluates P_1, w, q^2 with the polynomials, in double), so the
//     error of a coefficient grows by at most ~3 ulp per fade sample: <= 4 F 2^-53 relative at the end of a fade of F samples
//     (1e-12 for the longest fades of speech, F ~ 1500; tests/test_gpu_parity.py holds MODE_FAST to the usual bar, and fades of
//     350 000 samples in test_direct_stages_long_fades_hold_the_recurrence_bound).  Run without per-kind branches: a kind nobody
//     moves has the identity as its factor.  The anti-resonator N0 runs the recurrence on its plain pole pair and inverts per sample
//     (fast_anti_finish).  A GAIN advances by a constant increment (to - from) / F per fade sample from its exact first value (round 5;
//     the same bound: one rounding per sample), so only the source stage (the pitch) and N0 (whose frequency is TESTED for 0, reference
//     src/speechWaveGenerator.cpp:122, and must land on its target exactly) still form cnt / F;
//   * a fade START is a second masked block of 16-byte loads: klatt_seeds (below) has evaluated, densely and before the launch,
//     per frame and stage a RECORD -- what every kind needs to advance through the fade and the values of the fade's FIRST sample,
//     on which the reference re-evaluates everything (the previous fade's last interpolated value need not equal the frame value).
//     The state a record fills is kept in PAIRS of doubles that mirror the record's 16-byte entries (round 5), so that a masked
//     load lands in the very registers the sample loop reads: with one double per variable the compiler loaded into 44 temporaries
//     and copied them under the mask, which alone put a stage over the 128 registers of two workgroups per CU.  The 16-byte header
//     of the NEXT fade is loaded at the previous switch, as the flat stages do.
//
// Same arithmetic as every other kernel of the engine in MODE_EXACT (the seeds and the stages call the functions the tracks
// and the untracked stages call, on the same operands): the PCM is the same bytes (tests: tracked = direct = untracked).
//
// EIGHT stages, one wavefront each (klatt_device.h, direct_stage_kind): T0 source | T1 N0, NP | T2 r6, r5 | T3 r4, r3 | T4 r2, r1 |
// T5 frication, parallel 1, 2 | T6 parallel 3, 4 | T7 parallel 5, 6, mix, clip, PCM; which stage sits beside which on a SIMD is chosen
// from HW_ID (the pairs that balance the SIMDs differ between the modes).  MODE_FAST runs a second layout (klatt_device.h,
// direct_layout): the source in two stages (pitch / vibrato / phase | glottal wave, noise, gains), the cascade in two stages of three.
// TWO RESIDENCIES (template parameter WPE, wavefronts per SIMD): 2 -- one workgroup per CU, 16-sample hand-overs (152 KB of pipes),
// 256 registers per stage, round 4's kernel --, and 4 -- TWO workgroups per CU, 8-sample hand-overs (77 KB), 128 registers: the LEAN
// stages of round 5 (`a = 1 - b - c` derived where it is used instead of carried, the PCM tile flushed two rows at a time, shorter
// unrolls).  A launch of the unaligned batch is bound by the LATENCY of a sample through one workgroup's stage pipeline, which only a
// second resident workgroup overlaps.  DESIGN.md section 4.7 has the measurements.
#pragma once

#include "klatt_systolic.h"

namespace klatt {

// samples per trip of the mixed loops / of the steady loops, by arithmetic mode and residency (same PCM whatever the values)
#ifndef KLATT_DIRECT_UNROLL
#define KLATT_DIRECT_UNROLL 2            // MODE_EXACT, one workgroup per CU (4: 51.8 -> 57.9 ms on the all-different batch)
#endif
#ifndef KLATT_DIRECT_UNROLL_FAST
#define KLATT_DIRECT_UNROLL_FAST 4       // MODE_FAST, one workgroup per CU (2: 30.2 instead of 29.0 ms)
#endif
#ifndef KLATT_DIRECT_STEADY_UNROLL
#define KLATT_DIRECT_STEADY_UNROLL 16
#endif
#ifndef KLATT_DIRECT_LEAN_UNROLL
#define KLATT_DIRECT_LEAN_UNROLL 2       // MODE_EXACT, two workgroups per CU (1: 38.8 instead of 37.8 ms; the source stage goes sample by sample)
#endif
#ifndef KLATT_DIRECT_LEAN_UNROLL_FAST
#define KLATT_DIRECT_LEAN_UNROLL_FAST 8  // MODE_FAST, two workgroups per CU: a whole hand-over (4: 18.3 instead of 18.0 ms; before the stage pairs were re-balanced 2: 21.1, 1: 22.1, 4: 20.2)
#endif

// A pair of doubles in two adjacent register pairs: the unit a record entry is loaded in (global_load_dwordx4 into the state itself)
typedef double dpair __attribute__((ext_vector_type(2)));
__device__ __forceinline__ dpair make_dpair(double x, double y) { dpair v; v.x = x; v.y = y; return v; }

// ---- klatt_seeds: one record per (frame, stage) of the launch's direct utterances -------------------------------------------
// Record of stage s for one frame: direct_stage_entries(s) 16-byte entries,
//   resonator kind r (4 entries at 4 r):
//       MODE_EXACT                          (f_from, f_to - f_from) (bw_from, bw_to - bw_from) (b_1, c_1) (a_1, r_1)
//       MODE_FAST                           (Im P_1, Re w) (Im w, q^2) (b_1 = Re P_1, c_1 = -r_1^2) (a_1, -)
//       MODE_FAST, N0                       (Im P_1, Re w) (Im w, q^2) (Re P_1, r_1^2 of the PLAIN pole pair) (f_from, f_to - f_from)
//   gain kind g (3 entries at 4 NRES + 3 g) for its two parameters x, y:
//       MODE_EXACT                          (x_from, x_to - x_from) (y_from, y_to - y_from) (x_1, y_1)
//       MODE_FAST                           (x_1, y_1) ((x_to - x_from) / F, (y_to - y_from) / F) (-, -)
// where _1 is the value on the fade's first sample.  The end points follow reference src/frame.cpp:55-72 (host: DirectJob).
struct SeedArgs {
    const DirectJob* jobs;
    uint32_t nJobs;
    const double* frames;
    const FrameMeta* meta;
    DirectHdr* hdr;
    double2* rec;
    double negPiOverSr, twoPiOverSr;
};

template <int MODE, int S>
__device__ __forceinline__ void seed_stage(const SeedArgs& A, uint32_t j, bool mine)
{
    constexpr int LAY = direct_layout(MODE);
    constexpr int NR = direct_stage_res(S, LAY), NG = direct_stage_gains(S, LAY);
    const DirectJob job = A.jobs[j];
    const FrameMeta m = A.meta[job.frame];
    const double nf = (double)m.fadeSamples, invF = 1.0 / nf;
    const double ratio1 = div_by(1.0, nf, invF);
    auto value = [&](uint32_t fr, bool gate, int p) __attribute__((always_inline)) -> double {
        if (fr == kNoFrame || (p == 44 && gate)) return 0.0;
        return A.frames[(size_t)fr * kNumParams + p];
    };
    const bool gateFrom = job.flags & 1u, gateTo = job.flags & 2u;
    constexpr int kFirst = direct_stage_first(S, LAY), kEntries = direct_stage_entries(S, LAY);
    // A thread's record is kEntries x 16 bytes, so stores straight from the threads put 16 bytes into each of 64 lines per instruction
    // (1 TB/s for the 2.3 GB of a batch of 1.6 M frames).  The block stages its 256 records in LDS, entry-major with a pitch of 257
    // entries (consecutive entries of a record in different banks), and copies them out in the order they lie in memory.
    extern __shared__ __attribute__((aligned(16))) unsigned char seedLds[];
    double2* const stage = reinterpret_cast<double2*>(seedLds);
    constexpr int kPitch = 257;
    auto put = [&](int k, double2 v) __attribute__((always_inline)) { stage[k * kPitch + (int)threadIdx.x] = v; };
    uint32_t bits = 0;
#pragma unroll
    for (int r = 0; r < NR; ++r) {
        const int kind = direct_stage_kind(S, r, LAY);
        const double fF = value(job.from, gateFrom, kResF[kind]), fT = value(job.to, gateTo, kResF[kind]);
        const double bF = value(job.from, gateFrom, kResB[kind]), bT = value(job.to, gateTo, kResB[kind]);
        const double fd = fT - fF, bd = bT - bF;
        const bool anti = kind == 0;
        // the fade's first sample (what fade_update / klatt_tracks evaluate there)
        const double f1 = fF + (fd * ratio1), bw1 = bF + (bd * ratio1);
        const RadCos p1 = coefficient_parts(f1, bw1, A.negPiOverSr, A.twoPiOverSr);
        const Coef k1 = coefficient_finish(p1.rad, p1.cs, anti, f1);
        // what the fade moves; the classes of its arguments (fade_classes, klatt_systolic.h: the interpolated values stay between the
        // end points, with a margin against the one-ulp overshoot of from + (to - from) * ratio)
        const bool bwMoves = !(bT == bF), moves = bwMoves || !(fT == fF);
        const double xo = A.negPiOverSr * bF * kLog2e, xn = A.negPiOverSr * bT * kLog2e;
        const double to = A.twoPiOverSr * -fF * kTwoOverPi, tn = A.twoPiOverSr * -fT * kTwoOverPi;
        const bool eu = __builtin_fabs(xo) <= 0.499 && __builtin_fabs(xn) <= 0.499;
        const bool c0 = __builtin_fabs(to) <= 0.499 && __builtin_fabs(tn) <= 0.499;
        const bool c1 = to <= -0.501 && to >= -1.499 && tn <= -0.501 && tn >= -1.499;
        bits |= (moves ? 1u << r : 0u) | (bwMoves ? 1u << (kDirectBwShift + r) : 0u) |
                ((eu && c0) ? 0u : 1u << (kDirectClsShift + 2 * r)) | ((eu && c1) ? 0u : 2u << (kDirectClsShift + 2 * r));
        double2 e0, e1, e2, e3;
        if (MODE == MODE_FAST) {
            // P_1 = 2 r_1 e^(i theta_1); per fade sample theta advances by delta = (2 pi / sr) (-(f_to - f_from) / F) and r by the
            // factor q = exp((-pi / sr) ((bw_to - bw_from) / F)).  A fade of one sample never advances.
            const double th1 = A.twoPiOverSr * -f1;
            const double sn1 = fast_sin(th1);
            const double delta = A.twoPiOverSr * -(fd * invF), lq = A.negPiOverSr * (bd * invF);
            const bool one = m.fadeSamples <= 1u;
            const double q = one ? 1.0 : fast_exp(lq);
            e0 = make_double2(p1.rad * sn1 * 2.0, one ? 1.0 : q * fast_cos(delta));
            e1 = make_double2(one ? 0.0 : q * fast_sin(delta), q * q);
            e2 = make_double2(k1.b, k1.c);
            e3 = make_double2(k1.a, 0.0);
            if (anti) {
                // N0's coefficients are its pole pair's inverted (reference src/speechWaveGenerator.cpp:121-125): the recurrence runs on the
                // PLAIN pair (Re P = 2 r cos theta, r^2) and the stage inverts per sample; the frequency travels along for the `!= 0` test
                e2 = make_double2(p1.rad * p1.cs * 2.0, p1.rad * p1.rad);
                e3 = make_double2(fF, fd);
            }
        } else {
            e0 = make_double2(fF, fd);
            e1 = make_double2(bF, bd);
            e2 = make_double2(k1.b, k1.c);
            e3 = make_double2(k1.a, p1.rad);
        }
        put(4 * r, e0); put(4 * r + 1, e1); put(4 * r + 2, e2); put(4 * r + 3, e3);
    }
#pragma unroll
    for (int g = 0; g < NG; ++g) {
        const int kind = direct_stage_kind(S, NR + g, LAY);
        const int px = shape_param(entry_value(kind, 0)), py = entry_value(kind, 1) >= 0 ? shape_param(entry_value(kind, 1)) : -1;
        const double xF = value(job.from, gateFrom, px), xT = value(job.to, gateTo, px);
        const double yF = py >= 0 ? value(job.from, gateFrom, py) : 0.0, yT = py >= 0 ? value(job.to, gateTo, py) : 0.0;
        const double xd = xT - xF, yd = yT - yF;
        const bool moves = !(xT == xF) || !(yT == yF);
        bits |= moves ? 1u << (NR + g) : 0u;
        if (MODE == MODE_FAST) {
            put(4 * NR + 3 * g, make_double2(__builtin_fma(xd, ratio1, xF), __builtin_fma(yd, ratio1, yF)));
            put(4 * NR + 3 * g + 1, make_double2(xd * invF, yd * invF));
            put(4 * NR + 3 * g + 2, make_double2(0.0, 0.0));
        } else {
            put(4 * NR + 3 * g, make_double2(xF, xd));
            put(4 * NR + 3 * g + 1, make_double2(yF, yd));
            put(4 * NR + 3 * g + 2, make_double2(xF + (xd * ratio1), yF + (yd * ratio1)));
        }
    }
    __syncthreads();
    {
        const uint32_t first = blockIdx.x * 256u;                                  // the block's first frame
        const uint32_t nValid = A.nJobs - first < 256u ? A.nJobs - first : 256u;
        double2* const out = A.rec + (size_t)kFirst * A.nJobs + (size_t)first * kEntries;
        for (uint32_t i = threadIdx.x; i < nValid * (uint32_t)kEntries; i += 256u) {
            const uint32_t fr = i / (uint32_t)kEntries, k = i - fr * (uint32_t)kEntries;
            out[i] = stage[k * kPitch + fr];
        }
    }
    if (mine) {
        const unsigned long long mm = m.minSamples, ff = m.fadeSamples;
        const unsigned long long span = (mm > ff + 1ull ? mm : ff + 1ull) + 1ull;
        A.hdr[(size_t)S * A.nJobs + j] = DirectHdr{m.fadeSamples, (uint32_t)(span < 0xFFFFFFFFull ? span : 0xFFFFFFFFull), bits, 0u};
    }
}

// grid (ceil(nJobs / 256), 8): blockIdx.y is the stage, so the kinds a thread loops over are block-uniform and the wave-uniform
// short cuts of coefficient_parts ballot over full wavefronts (the last block repeats the last job in its idle lanes)
template <int MODE>
__global__ void __launch_bounds__(256) klatt_seeds(const SeedArgs A)
{
    const uint32_t jj = blockIdx.x * 256u + threadIdx.x;
    const bool mine = jj < A.nJobs;
    const uint32_t j = mine ? jj : A.nJobs - 1u;
    switch (blockIdx.y) {
    case 0: seed_stage<MODE, 0>(A, j, mine); break;
    case 1: seed_stage<MODE, 1>(A, j, mine); break;
    case 2: seed_stage<MODE, 2>(A, j, mine); break;
    case 3: seed_stage<MODE, 3>(A, j, mine); break;
    case 4: seed_stage<MODE, 4>(A, j, mine); break;
    case 5: seed_stage<MODE, 5>(A, j, mine); break;
    case 6: seed_stage<MODE, 6>(A, j, mine); break;
    default: seed_stage<MODE, 7>(A, j, mine); break;
    }
}

// ---- a direct stage's state ----------------------------------------------------------------------------------------------------
template <int STAGE_, int LAY_ = 0>
struct DirectDesc {
    static constexpr int NST = kDirectStages, STAGE = STAGE_, LAYOUT = LAY_;
    static constexpr int NRES = direct_stage_res(STAGE_, LAY_), NGAIN = direct_stage_gains(STAGE_, LAY_), NE = NRES + NGAIN;
    static constexpr bool ANTI0 = direct_stage_kind(STAGE_, 0, LAY_) == 0;      // the stage's first resonator is N0
    static constexpr int ENTRIES = direct_stage_entries(STAGE_, LAY_);
};
// LEAN: the 128-register stages of two workgroups per CU -- `a` of a plain resonator is derived where it is used, not carried
template <class DD, int MODE, bool LEAN>
struct DirectState {
    static constexpr int NR = DD::NRES > 0 ? DD::NRES : 1, NG = DD::NGAIN > 0 ? DD::NGAIN : 1;
    static constexpr bool FASTANTI = MODE == MODE_FAST && DD::ANTI0;          // resonator 0 is N0 on the recurrence of its plain pole pair
    // whether cnt / F is formed per fade sample: MODE_EXACT interpolates with it; in MODE_FAST only the pitch (stage 0) and N0's frequency
    static constexpr bool RATIO = MODE != MODE_FAST || DD::STAGE == 0 || DD::ANTI0;
    // resonator r, in the pairs of its record: bc = (b, c) [N0 in MODE_FAST: (Re P, r^2) of the plain pole pair];
    // MODE_EXACT e0 = (f_from, f_delta), e1 = (bw_from, bw_delta), rad = r;  MODE_FAST e0 = (Im P, Re w), e1 = (Im w, q^2)
    dpair bc[NR], e0[NR], e1[NR];
    double ra[NR];                                      // a: carried by the stages of one workgroup per CU, and for N0 (whose a is not 1 - b - c)
    double rad[NR];
    double z1[NR], z2[NR];
    dpair fn;                                           // MODE_FAST, N0: its frequency's (from, to - from)
    double n0b, n0c;                                    // MODE_FAST, N0: the inverted pair's b, c
    // gain kind g: gv = the current (x, y); MODE_EXACT gx = (x_from, x_delta), gy = (y_from, y_delta); MODE_FAST gx = the increments of (x, y)
    dpair gv[NG], gx[NG], gy[NG];
    double invF;                                        // 1 / F
    double ratio;                                       // source stage only: cnt / F of the values in registers (the pitch fades with them)
    bool inFade;                                        // source stage only: the values in registers belong to a fade sample
    uint32_t cnt, F;                                    // the fade-sample index of the values in registers; cnt == F: not fading
    uint32_t startAt, next, nFrames, length;            // startAt: the sample the next fade's first values apply to
    uint32_t curBits;                                   // DirectHdr.bits of the running (or last) fade
    uint32_t rec0;                                      // the record number of the utterance's first frame
    DirectHdr nextHdr;                                  // frame `next`'s, loaded ahead
    bool live;
    static __device__ __forceinline__ constexpr bool keepsA(int r) { return !LEAN || (DD::ANTI0 && r == 0); }
    __device__ __forceinline__ double a(int r) const { return keepsA(r) ? ra[r] : (1.0 - bc[r].x) - bc[r].y; }      // reference src/speechWaveGenerator.cpp:119
    __device__ __forceinline__ double b(int r) const { return (FASTANTI && r == 0) ? n0b : bc[r].x; }
    __device__ __forceinline__ double c(int r) const { return (FASTANTI && r == 0) ? n0c : bc[r].y; }
    // One sample of resonator r (reference src/speechWaveGenerator.cpp:128-135).  MODE_EXACT: ((a in) + (b z1)) + (c z2), each operation
    // rounded by itself.  The lean stages of MODE_FAST, whose a IS 1 - b - c, take in + b (z1 - in) + c (z2 - in): the same value with
    // four operations instead of five (two for a), and no product with a, the small difference of large terms.
    __device__ __forceinline__ double step(int r, double in)
    {
        double y;
        if (MODE == MODE_FAST && !keepsA(r)) y = __builtin_fma(bc[r].y, z2[r] - in, __builtin_fma(bc[r].x, z1[r] - in, in));
        else y = dot3<MODE>(a(r), in, b(r), z1[r], c(r), z2[r]);
        z2[r] = z1[r]; z1[r] = y;
        return y;
    }
};
struct DirectCtx {
    const KernelArgs& A;
    const DirectHdr* hdr;      // the stage's headers [nDirect]
    const dpair* rec;          // the stage's records
};
template <class DD, int MODE, bool LEAN>
__device__ __forceinline__ void direct_init(DirectState<DD, MODE, LEAN>& f, bool live, const UttDesc& d, uint32_t rec0, const DirectCtx& X)
{
#pragma unroll
    for (int r = 0; r < DD::NRES; ++r) {
        f.ra[r] = 0; f.bc[r] = make_dpair(2.0, -1.0); f.z1[r] = 0; f.z2[r] = 0;      // the coefficients of f = bw = 0
        f.e0[r] = make_dpair(0.0, 0.0); f.e1[r] = make_dpair(0.0, 0.0); f.rad[r] = 1.0;
    }
    if (DirectState<DD, MODE, LEAN>::FASTANTI) f.bc[0] = make_dpair(2.0, 1.0);          // (Re P, r^2) of the plain pair
    f.fn = make_dpair(0.0, 0.0); f.n0b = 2.0; f.n0c = -1.0;
#pragma unroll
    for (int g = 0; g < DD::NGAIN; ++g) { f.gv[g] = make_dpair(0.0, 0.0); f.gx[g] = make_dpair(0.0, 0.0); f.gy[g] = make_dpair(0.0, 0.0); }
    f.live = live && d.length > 0u;
    f.nFrames = d.nFrames; f.length = d.length; f.next = 0;
    f.cnt = 1u; f.F = 1u; f.invF = 1.0; f.ratio = 1.0; f.inFade = false; f.curBits = 0u; f.rec0 = rec0;
    const bool any = live && d.nFrames > 0u;
    f.startAt = any ? 1u : 0xFFFFFFFFu;      // frame 0 is dequeued on sample 0, its fade's first values apply to sample 1
    f.nextHdr = DirectHdr{1u, 0u, 0u, 0u};
    if (any) f.nextHdr = X.hdr[rec0];
}
// OR over the wavefront, as a scalar
__device__ __forceinline__ uint32_t wave_or(uint32_t v)
{
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v |= (uint32_t)__shfl_xor((int)v, m, kLanes);
    return (uint32_t)__builtin_amdgcn_readfirstlane((int)v);
}

// MODE_FAST, N0: the anti-resonator's coefficients from its plain pole pair (Re P = 2 r cos theta, r^2) and its frequency (reference
// src/speechWaveGenerator.cpp:119-125: a = 1 - b - c, and for a frequency other than 0 the inversion a' = 1 / a, b' = -b a', c' = -c a').
// The reciprocal is v_rcp_f64 with one Newton step.  A relative error e of (b, c) arrives in the coefficients as ~e (|b| + |c|) / a:
// the recurrence's 4 F 2^-53 becomes <= 12 F 2^-53 / a with a = |1 - r e^(i theta)|^2 (5e-3 for a nasal zero at 300 Hz, 100 Hz wide).
__device__ __forceinline__ void fast_anti_finish(double reP, double r2, double fr, double& ra, double& rb, double& rc)
{
    const double b = reP, c = -r2;
    const double a0 = (1.0 - b) - c;
    double inv = __builtin_amdgcn_rcp(a0);
    inv = __builtin_fma(__builtin_fma(-a0, inv, 1.0), inv, inv);
    const bool nz = fr != 0.0;
    ra = nz ? inv : a0;
    rb = nz ? b * -inv : b;
    rc = nz ? c * -inv : c;
}

// Called by a stage's sample BETWEEN its two halves (the flat stages' FlatMid): the first half has read every coefficient and gain
// of the sample; here every lane advances its fade-sample index and evaluates, for the NEXT sample, the kinds the chunk's mask
// `wm` names; then the lanes whose next fade's first values apply to the next sample (`sw`) switch: everything of the new fade
// comes from its record, in one block of loads into the state's own registers.
template <class DD, int MODE, bool LEAN>
struct DirectMid {
    DirectState<DD, MODE, LEAN>& f;
    const DirectCtx& X;
    uint32_t wm;       // wave-uniform (SGPR): DirectHdr.bits OR-ed over the lanes that fade or start a fade in this chunk
    bool sw;
    // `token`: the last value the sample computed.  The fade-sample counter is tied to it below (an empty asm statement that "modifies"
    // the counter with the token as input), so that the block that overwrites the stage's coefficients and gains cannot be scheduled
    // ABOVE the arithmetic that reads them -- the compiler did exactly that (the block depends on nothing the sample computes), and
    // then kept a copy of every value for the lanes the block masks off: six to eight v_mov_b64 per stage and sample.
    __device__ __forceinline__ void operator()(double token) const
    {
        using ST = DirectState<DD, MODE, LEAN>;
        const KernelArgs& A = X.A;
        asm volatile("" : "+v"(f.cnt) : "v"(token));
        // ONE masked block for the lanes inside a fade (a lane that is not keeps what it has: the fade's last values, as the reference
        // does until the next fade's first sample); skipped altogether on a sample on which no lane of the wavefront fades
        const bool adv = f.cnt < f.F;
        if (DD::STAGE == 0) f.inFade = adv || sw;
        if (adv) {
            const uint32_t cn = f.cnt + 1u;
            f.cnt = cn;
            // ratio = (double)counter / numFadeSamples, correctly rounded (reference src/frame.cpp:49) -- in MODE_FAST too: the last sample
            // of a fade must land on its target EXACTLY where the reference tests a parameter for a value (`frequency != 0` decides whether
            // N0 is inverted, src/speechWaveGenerator.cpp:122: with counter * (1 / F) = 1 - 1e-16 a target of 0 Hz arrives as 1e-14 Hz)
            double ratio = 0.0;
            if (ST::RATIO) ratio = div_by((double)cn, (double)f.F, f.invF);
            if (DD::STAGE == 0) f.ratio = ratio;
#pragma unroll
            for (int r = 0; r < DD::NRES; ++r) {
                const bool anti = DD::ANTI0 && r == 0;
                // (MODE_FAST's recurrences are seven instructions: cheaper to run for a kind nobody moves -- its factor is the identity --
                // than to branch around; the polynomials are worth a scalar branch)
                if (MODE != MODE_FAST && !(wm & (1u << r))) continue;
                if (MODE == MODE_FAST) {
                    // P <- P w, r^2 <- r^2 q^2: bc.x = Re P, e0.x = Im P, (e0.y, e1.x) = w, bc.y = -r^2 (N0: r^2), e1.y = q^2
                    const double re = __builtin_fma(-f.e0[r].x, f.e1[r].x, f.bc[r].x * f.e0[r].y);
                    const double im = __builtin_fma(f.bc[r].x, f.e1[r].x, f.e0[r].x * f.e0[r].y);
                    const double c = f.bc[r].y * f.e1[r].y;
                    f.bc[r].x = re; f.e0[r].x = im; f.bc[r].y = c;
                    if (anti) fast_anti_finish(re, c, __builtin_fma(f.fn.y, ratio, f.fn.x), f.ra[r], f.n0b, f.n0c);      // the plain pair advanced like any other; then the inversion
                    else if (ST::keepsA(r)) f.ra[r] = (1.0 - re) - c;
                } else {
                    const uint32_t cls = (wm >> (kDirectClsShift + 2 * r)) & 3u;
                    const double fr = f.e0[r].x + (f.e0[r].y * ratio);      // reference src/utils.h:22
                    const double th = A.twoPiOverSr * -fr;
                    const double cs = !(cls & 1u) ? cos_unreduced(th) : (!(cls & 2u) ? cos_quadrant_m1(th) : fast_cos(th));      // (constants as scalar operands: klatt_math.h)
                    if (wm & (1u << (kDirectBwShift + r))) {
                        const double bw = f.e1[r].x + (f.e1[r].y * ratio);
                        const double ex = A.negPiOverSr * bw;
                        f.rad[r] = cls != 3u ? exp_unreduced(ex) : fast_exp(ex);
                    }
                    const Coef k = coefficient_finish(f.rad[r], cs, anti, fr);      // reference src/speechWaveGenerator.cpp:117-126
                    f.bc[r].x = k.b; f.bc[r].y = k.c;
                    if (ST::keepsA(r)) f.ra[r] = k.a;
                }
            }
#pragma unroll
            for (int g = 0; g < DD::NGAIN; ++g) {
                if (MODE == MODE_FAST) {
                    f.gv[g] = f.gv[g] + f.gx[g];
                } else {
                    if (!(wm & (1u << (DD::NRES + g)))) continue;
                    f.gv[g].x = f.gx[g].x + (f.gx[g].y * ratio);
                    f.gv[g].y = f.gy[g].x + (f.gy[g].y * ratio);
                }
            }
        }
        if (sw) {
            const DirectHdr h = f.nextHdr;
            const dpair* const rec = X.rec + (size_t)(f.rec0 + f.next) * DD::ENTRIES;
            const double* const recD = reinterpret_cast<const double*>(rec);
#pragma unroll
            for (int r = 0; r < DD::NRES; ++r) {
                f.e0[r] = rec[4 * r]; f.e1[r] = rec[4 * r + 1]; f.bc[r] = rec[4 * r + 2];
                if (ST::FASTANTI && r == 0) f.fn = rec[4 * r + 3];
                else {
                    if (ST::keepsA(r)) f.ra[r] = recD[2 * (4 * r + 3)];
                    if (MODE != MODE_FAST) f.rad[r] = recD[2 * (4 * r + 3) + 1];
                }
            }
#pragma unroll
            for (int g = 0; g < DD::NGAIN; ++g) {
                const int at = 4 * DD::NRES + 3 * g;
                if (MODE == MODE_FAST) { f.gv[g] = rec[at]; f.gx[g] = rec[at + 1]; }
                else { f.gx[g] = rec[at]; f.gy[g] = rec[at + 1]; f.gv[g] = rec[at + 2]; }
            }
            f.cnt = 1u; f.F = h.fadeSamples; f.curBits = h.bits;
            const double nfD = (double)h.fadeSamples;
            if (ST::RATIO) f.invF = 1.0 / nfD;
            if (DD::STAGE == 0) f.ratio = div_by(1.0, nfD, f.invF);
            f.next++;
            const bool more = f.next < f.nFrames;
            f.startAt = more ? f.startAt + h.span : 0xFFFFFFFFu;
            f.nextHdr = X.hdr[f.rec0 + (more ? f.next : f.nFrames - 1u)];      // (past the last frame: any valid header; never used)
            // Everything loaded is waited for HERE, inside the block that only switching lanes enter: left to the compiler, the waits
            // sit at the first uses -- on every sample of the loop.
#pragma unroll
            for (int r = 0; r < DD::NRES; ++r) {
                flat_pin(f.e0[r]); flat_pin(f.e1[r]); flat_pin(f.bc[r]);
                if (ST::keepsA(r) && !(ST::FASTANTI && r == 0)) flat_pin(f.ra[r]);
                if (MODE != MODE_FAST) flat_pin(f.rad[r]);
            }
            if (ST::FASTANTI) {
                // (klatt_seeds: Re P_1, r_1^2, and the frequency's end points; the first sample's coefficients as every later one's)
                flat_pin(f.fn);
                fast_anti_finish(f.bc[0].x, f.bc[0].y, __builtin_fma(f.fn.y, div_by(1.0, nfD, f.invF), f.fn.x), f.ra[0], f.n0b, f.n0c);
            }
#pragma unroll
            for (int g = 0; g < DD::NGAIN; ++g) { flat_pin(f.gv[g]); flat_pin(f.gx[g]); if (MODE != MODE_FAST) flat_pin(f.gy[g]); }
            flat_pin(f.nextHdr.fadeSamples); flat_pin(f.nextHdr.span); flat_pin(f.nextHdr.bits);
        }
    }
};

// The chunk's mask: what the lanes that fade, or start a fade, in samples t0 + 1 .. t1 + 1 move, and the classes of their arguments.
// A lane whose next TWO fades start inside the chunk (frames of a few samples) asks for everything.
template <class ST>
__device__ __forceinline__ uint32_t direct_chunk_mask(const ST& f, uint32_t t1)
{
    uint32_t lb = 0;
    if (f.cnt < f.F) lb |= f.curBits;
    if (f.startAt <= t1) {
        lb |= f.nextHdr.bits;
        if (f.nextHdr.span <= t1 - f.startAt) lb |= kDirectAllBits;
    }
    return wave_or(lb);
}

// load(c, i): the stage's pipe inputs of sample i (a struct of doubles); body(c, i, in, mid): one sample of the stage.  In a mixed chunk
// the inputs of sample i + 1 are read while sample i is computed (an LDS read issued at the top of a sample is waited for right there,
// in front of the sample's dependent chain; the read past the chunk's last sample lands in the pipe's other buffer and is dropped).
// Barrier discipline of flat2_loop (klatt_systolic.h).
struct In0 {};
struct In1 { double a; };
struct In2 { double a, b; };
struct In3 { double a, b, c; };
template <int MODE, bool LEAN> struct DirectUnroll {
    static constexpr int kMixed = LEAN ? (MODE == MODE_FAST ? KLATT_DIRECT_LEAN_UNROLL_FAST : KLATT_DIRECT_LEAN_UNROLL) : (MODE == MODE_FAST ? KLATT_DIRECT_UNROLL_FAST : KLATT_DIRECT_UNROLL);
    static constexpr int kSteady = KLATT_DIRECT_STEADY_UNROLL;      // (the lean stages have no steady loop of their own)
};
template <class DD, int MODE, int CH, bool LEAN, class FLoad, class FBody, class FChunk>
__device__ __forceinline__ void direct_loop(int depth, int nIter, int nChunks, int stampSlot, DirectState<DD, MODE, LEAN>& f, const DirectCtx& X, FLoad load, FBody body, FChunk perChunk)
{
    constexpr int kMixedUnroll = DirectUnroll<MODE, LEAN>::kMixed, kSteadyUnroll = DirectUnroll<MODE, LEAN>::kSteady;
#ifdef KLATT_STAMPS
    Stamps st;
#endif
    for (int iter = 0; iter < nIter; ++iter) {
        STAMP_BEGIN();
        STAMP_IDLE();
        const int c = iter - depth;
        if (c >= 0 && c < nChunks) {
            const uint32_t t0 = (uint32_t)c * (uint32_t)CH, t1 = t0 + (uint32_t)CH;
            if (f.length <= t0) f.live = false;                               // this lane has emitted its last sample
            // a chunk is steady when no lane evaluates anything in it: none fading, no fade whose first values apply to t0 + 1 .. t1
            const bool busy = f.cnt < f.F || f.startAt <= t1;
            if (!LEAN && !__any(busy)) {
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
                for (uint32_t q = 0; q < run; ++q) {
                    if (f.live) {
#pragma unroll kSteadyUnroll
                        for (int i = 0; i < CH; ++i) body(cc, i, load(cc, i), NoMid{});
                    }
                    perChunk((uint32_t)(cc + 1) * (uint32_t)CH);
                    if (q + 1 < run) {
                        STAMP_WORKED();
                        __syncthreads();
                        STAMP_SYNCED();
                        STAMP_BEGIN();
                        ++iter; ++cc;
                    }
                }
            } else {
                STAMP_KIND(-1);
                const uint32_t wm = direct_chunk_mask(f, t1);
                auto ahead = load(c, 0);
#pragma unroll kMixedUnroll
                for (int i = 0; i < CH; ++i) {
                    const bool sw = t0 + (uint32_t)i + 1u == f.startAt;
                    const auto in = ahead;
                    ahead = load(c, i + 1);
                    body(c, i, in, DirectMid<DD, MODE, LEAN>{f, X, wm, sw});
                }
                perChunk(t1);
            }
        }
        STAMP_WORKED();
        __syncthreads();
        STAMP_SYNCED();
    }
#ifdef KLATT_STAMPS
    if (X.A.debug && (threadIdx.x & (kLanes - 1)) == 0) {
        unsigned long long* o = X.A.debug + (blockIdx.x * DD::NST + stampSlot) * 8;
        o[0] = st.work; o[1] = st.wait; o[2] = st.n[0]; o[3] = st.n[1]; o[4] = st.n[2]; o[5] = st.c[0]; o[6] = st.c[1]; o[7] = st.c[2];
    }
#endif
    (void)stampSlot;
}


// sin() of the vibrato (reference src/speechWaveGenerator.cpp:77), out of line: inlined, the device library's sine brings its two dozen
// constants and its temporaries into the source stage's sample loop, which runs it for the rare utterance with vibrato only
__device__ __attribute__((noinline)) double direct_vib_sin(double x) { return sin(x); }

// ---- the source stage (T0) -------------------------------------------------------------------------------------------------------
// The pitch glides with the sample count of THIS utterance (reference src/frame.cpp:76-79, :98, :71): per sample a lane is
// dequeuing (sets up the pitch fade; the sample itself is emitted unchanged), fading (pitch interpolated), ending its fade
// or steady (glide) -- the flat source stage's selects (klatt_systolic.h).  What a dequeue reads (the frame's two pitch values,
// index mark) is a SourceRef loaded when the previous frame was dequeued.  Gains: gv[0] = (vibratoPitchOffset, vibratoSpeed),
// and in layout 0 gv[1] = (turbulence, openQuotient), gv[2] = (voiceAmplitude, aspirationAmplitude), gv[3] = (preFormantGain, -).
// PHASE (layout 1, MODE_FAST): the stage ends with the pitch phase, which it hands to the glottal stage -- its only parameters are the
// vibrato's.
template <class DD, int MODE, int CH, bool LEAN, bool PHASE>
__device__ __forceinline__ void direct_source_stage(const KernelArgs& A, const UttDesc& d, bool live, uint32_t u, uint32_t rec0, const DirectCtx& X, int lane,
                                                    int nIter, int nChunks, double* pipeOut, uint32_t nkey, uint32_t ninc, uint32_t ninc2)
{
#define SRC_PIPE(c, i) pipeOut[(((c) & 1) * CH + (i)) * kLanes + lane]
    // (the lean source stage of MODE_EXACT -- four gain kinds, the pitch, the phases -- goes sample by sample: two per trip spill)
    constexpr int kMixedUnroll = (LEAN && MODE != MODE_FAST) ? 1 : DirectUnroll<MODE, LEAN>::kMixed, kSteadyUnroll = DirectUnroll<MODE, LEAN>::kSteady;
    DirectState<DD, MODE, LEAN> f;
    direct_init(f, live, d, rec0, X);
    const SourceRef* const mySrc = A.sourceRef + d.frameStart;
    PitchState ps;
    ps.cur0 = 0.0; ps.old0 = 0.0; ps.new0 = 0.0; ps.oldInc = 0.0; ps.newInc = 0.0;
    double pitchPhase = 0.0, vibPhase = 0.0, aspNoise = 0.0;
    uint32_t noiseSt = noise_first(nkey, ninc);    // aspiration: noise values 0, 2, 4, ...
    bool wasFade = false;                          // the previous sample was a fade sample
    int32_t lastIndex = -1;
    bool oldNull = true, newNull = false;
    SourceRef nextSrc{0.0, 0.0, 1.0, -1, 0u};     // frame `f.next`, loaded ahead like f.nextHdr
    if (live && d.nFrames > 0u) nextSrc = mySrc[0];
    auto source = [&](bool waveVib, const auto& mid) __attribute__((always_inline)) -> double {
        double vib = 1.0;
        if (waveVib) {
            const double vs = f.gv[0].y;
            const double adv = frac_toward_zero(div_by(vs, A.sampleRateF, A.invSampleRate) + vibPhase);
            vibPhase = (vs != 0.0) ? adv : vibPhase;
            vib = (direct_vib_sin(vibPhase * 6.283185307179586) * 0.06 * f.gv[0].x) + 1.0;
        }
        pitchPhase = frac_toward_zero(div_by(ps.cur0 * vib, A.sampleRateF, A.invSampleRate) + pitchPhase);
        if constexpr (PHASE) {
            mid(pitchPhase);
            return pitchPhase;
        } else {
            const double turbGain = f.gv[1].x, openQ = f.gv[1].y, voiceAmp = f.gv[2].x, aspAmp = f.gv[2].y, preGain = f.gv[3].x;
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
            mid(out);      // the next sample's values: every parameter of this one has been used
            return out;
        }
    };
    auto vib_live = [&]() __attribute__((always_inline)) -> bool { return f.gv[0].x != 0.0 || f.gv[0].y != 0.0 || vibPhase != vibPhase; };
#ifdef KLATT_STAMPS
    Stamps st;
#endif
    for (int iter = 0; iter < nIter; ++iter) {
        STAMP_BEGIN();
        STAMP_IDLE();
        const int c = iter;
        if (c < nChunks) {
            const uint32_t t0 = (uint32_t)c * (uint32_t)CH, t1 = t0 + (uint32_t)CH;
            if (f.length <= t0) f.live = false;
            const bool busy = f.cnt < f.F || f.startAt <= t1 || f.inFade || wasFade || vib_live();
            STAMP_KIND(__any(busy) ? -1 : 0);
            if (!LEAN && !__any(busy)) {
                // steady stretch, decided once: the pitch glides, nothing else changes
                uint32_t run = f.live ? (f.startAt - t0 - 1u) / (uint32_t)CH : 0xFFFFFFFFu;
#pragma unroll
                for (int m = 32; m >= 1; m >>= 1) { const uint32_t o = (uint32_t)__shfl_xor((int)run, m, kLanes); run = o < run ? o : run; }
                run = (uint32_t)__builtin_amdgcn_readfirstlane((int)run);
                const uint32_t room = (uint32_t)(nChunks - c);
                run = run < room ? run : room;
                run = run < 1u ? 1u : run;
                int cc = c;
                for (uint32_t q = 0; q < run; ++q) {
                    if (f.live) {
#pragma unroll kSteadyUnroll
                        for (int i = 0; i < CH; ++i) { ps.cur0 += ps.oldInc; SRC_PIPE(cc, i) = source(false, NoMid{}); }
                        ps.old0 = ps.cur0;
                    }
                    if (q + 1 < run) { STAMP_WORKED(); __syncthreads(); STAMP_SYNCED(); STAMP_BEGIN(); ++iter; ++cc; }
                }
            } else {
                const uint32_t wm = direct_chunk_mask(f, t1);
                // vibrato can only come alive in this chunk through its kind (the phase only turns NaN while it advances)
                const bool vibChunk = (wm & (1u << DD::NRES)) != 0u || __any(vib_live());
#pragma unroll kMixedUnroll
                for (int i = 0; i < CH; ++i) {
                    const uint32_t t = t0 + (uint32_t)i;
                    const bool deq = t + 1u == f.startAt;
                    if (deq) {   // reference src/frame.cpp:55-72; the sample itself is emitted as it is
                        const SourceRef m = nextSrc;
                        newNull = (m.flags & FRAME_NULL) != 0;
                        ps.new0 = newNull ? ps.cur0 : m.pitch;
                        ps.newInc = newNull ? 0.0 : m.pitchInc;                // reference src/frame.cpp:98 (the division: host)
                        if (!newNull && oldNull) ps.old0 = m.pitch;
                        oldNull = newNull;                                    // for the NEXT dequeue: this fade has ended by then (:44-47)
                        if (m.userIndex != -1) lastIndex = m.userIndex;       // (:69)
                        ps.new0 += ps.newInc * (double)f.nextHdr.fadeSamples; // (:71)
                        nextSrc = mySrc[f.next + 1u < f.nFrames ? f.next + 1u : f.nFrames - 1u];
                    }
                    // The pitch of this sample, as selects.  A fade sample (the stage's values in registers are a fade's: f.inFade, set
                    // by the previous sample's DirectMid together with f.ratio = cnt / F) -> interpolated; the sample after a fade's last ->
                    // the fade's target becomes the glide's start; a steady sample -> glide; a dequeuing lane leaves it alone
                    // (reference src/frame.cpp:48-53, :44-47, :76-79, :55-72).  A fade is at least one sample, a frame at least two more:
                    // the four cases never coincide.
                    const bool fad = f.inFade;
                    const bool ending = !fad && wasFade;
                    const bool glide = !fad && !ending && !deq;
                    const double fv = fade_value(ps.old0, ps.new0, f.ratio);
                    const double gv = ps.cur0 + ps.oldInc;
                    ps.cur0 = fad ? fv : (glide ? gv : ps.cur0);
                    ps.old0 = ending ? ps.new0 : (glide ? gv : ps.old0);
                    ps.oldInc = ending ? ps.newInc : ps.oldInc;
                    wasFade = fad;
                    const bool waveVib = vibChunk && __any(vib_live());
                    SRC_PIPE(c, i) = source(waveVib, DirectMid<DD, MODE, LEAN>{f, X, wm, deq});
                }
            }
        }
        STAMP_WORKED();
        __syncthreads();
        STAMP_SYNCED();
    }
#ifdef KLATT_STAMPS
    if (A.debug && lane == 0) {
        unsigned long long* o = A.debug + (blockIdx.x * DD::NST + 0) * 8;
        o[0] = st.work; o[1] = st.wait; o[2] = st.n[0]; o[3] = st.n[1]; o[4] = st.n[2]; o[5] = st.c[0]; o[6] = st.c[1]; o[7] = st.c[2];
    }
#endif
    if (live) {
        UttResult res;
        res.produced = d.length; res.framesTaken = f.next; res.lastIndex = lastIndex; res.drained = 1u;
        A.result[u] = res;
    }
#undef SRC_PIPE
}

// ---- the kernel: 8 wavefronts = 8 direct stages over the same 64 utterances ----------------------------------------------------
// Pipes [2 buffers][CH][64 lanes] f64, one workgroup barrier per chunk, a stage of depth d on chunk iter - d (klatt_systolic.h):
//   T0 --x0--> T1 --x1--> T2 --x2--> T3 --x3--> T4 --o--> T7          depths 0 1 2 3 4 . . 5
//   T5 --(y, part)--> T6 --(y, part)--> T7                            depths 3 4 5   (T5 has no input: it starts three chunks late)
template <int CH>
struct DirectLds {
    static constexpr int kPipeBytes = 2 * CH * kLanes * 8;
    static constexpr int kNumPipes = 9;
    static constexpr int kTileOff = kNumPipes * kPipeBytes;
    static constexpr int kRowBase = kTileOff + kLanes * kTileStride;
    static constexpr int kRowCount = kRowBase + kLanes * 8;
    static constexpr int kMaxLen = kRowCount + kLanes * 4;
    static constexpr int kBytes = kMaxLen + 32;      // (longest utterance; waves per SIMD) CH = 16: 152 864 B, one workgroup per CU; CH = 8: 79 136 B, two
};
// WPE: wavefronts per SIMD the register budget is set for (2: one workgroup per CU, 256 VGPRs; 4: two, 128: the lean stages).
// ONLY: the stages whose bodies are compiled in (register census, tools/direct_census.sh; bit 0 T0, 1 the nasal pair, 2 the cascade,
// 3 T5, 4 T6, 5 T7, 6 layout 1's glottal stage); the library instantiates all of them.
constexpr int kDirectAllStages = 0x7F;
template <int MODE, int CH, int WPE, int ONLY = kDirectAllStages>
__global__ void __launch_bounds__(kLanes * 8, WPE) klatt_direct(const KernelArgs A)
{
    using L = DirectLds<CH>;
    constexpr int kChunk = CH;
    constexpr bool LEAN = WPE >= 4;
    constexpr int LAY = direct_layout(MODE);      // which stage runs what (klatt_device.h)
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    auto pipe = [&](int k) __attribute__((always_inline)) { return reinterpret_cast<double*>(lds + k * L::kPipeBytes); };
    double* const pipeX0 = pipe(0); double* const pipeX1 = pipe(1); double* const pipeX2 = pipe(2); double* const pipeX3 = pipe(3);
    double* const pipeO = pipe(4); double* const pipeY = pipe(5); double* const pipeP = pipe(6); double* const pipeY2 = pipe(7); double* const pipeP2 = pipe(8);
    unsigned char* const tile = lds + L::kTileOff;
    long long* const rowBase = reinterpret_cast<long long*>(lds + L::kRowBase);
    uint32_t* const rowCount = reinterpret_cast<uint32_t*>(lds + L::kRowCount);
    uint32_t* const maxLenP = reinterpret_cast<uint32_t*>(lds + L::kMaxLen);

    const int lane = threadIdx.x & (kLanes - 1);
    const long long slot = (long long)blockIdx.x * kLanes + lane;
    const uint32_t u = (slot < A.nSlots) ? A.order[slot] : 0xFFFFFFFFu;
    const bool live = (u != 0xFFFFFFFFu);

    UttDesc d;
    d.frameStart = 0; d.outStart = 0; d.nFrames = 0; d.seed = 0; d.flags = 0; d.length = 0;
    uint32_t rec0 = 0;
    if (live) { d = A.utt[u]; rec0 = A.directFirst[u]; }
    const uint32_t nkey = noise_key(d.seed), ninc = noise_inc(d.seed), ninc2 = noise_inc2(ninc);
    constexpr int FINAL = 7;

    // Which wave runs which stage.  The eight waves of the workgroup sit two to a SIMD, and a SIMD issues for one of them at a time:
    // what bounds a sample-step is the pair of stages with the most instructions between them.  So a wave takes its stage from the
    // SIMD it finds itself on (HW_REG_HW_ID) and its rank among the workgroup's waves there, pairing a heavy stage with a light one
    // (same-run A/B of the candidates: profiles/r4_direct_ab.txt); if the hardware placed the waves otherwise than two per SIMD,
    // stage = wave index (any bijection is correct -- the waves are interchangeable until they pick a stage).
    uint32_t* const simdCount = maxLenP + 4;
    if (threadIdx.x < 5) maxLenP[threadIdx.x < 1 ? 0 : threadIdx.x + 3] = 0;
    __syncthreads();
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    int stage = wave;
    {
        const uint32_t hw = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));   // HW_REG_HW_ID: wave slot [3:0], SIMD [5:4]
        const uint32_t simd = (hw >> 4) & 3u;
        uint32_t rank = 0;
        if (lane == 0) rank = atomicAdd(&simdCount[simd], 1u);
        rank = (uint32_t)__builtin_amdgcn_readfirstlane((int)rank);
        __syncthreads();
        const bool two = simdCount[0] == 2u && simdCount[1] == 2u && simdCount[2] == 2u && simdCount[3] == 2u;
        // heavy with light (same-run A/B of the candidates at two workgroups per CU: profiles/r5_direct_ab.txt):
        // MODE_EXACT T5 + T3 | T7 + T4 | T0 + T2 | T6 + T1;  MODE_FAST (layout 1): nasal pair + glottal | phase + parallel 3, 4 |
        // final + first cascade stage | parallel 1, 2 + second cascade stage
        constexpr int kPairsExact[8] = {5, 3, 7, 4, 0, 2, 6, 1}, kPairsFast0[8] = {1, 2, 0, 3, 5, 4, 7, 6}, kPairsFast1[8] = {2, 1, 0, 6, 7, 3, 5, 4};
        const int key = (int)(simd * 2u + (rank & 1u));
        int pick = wave;
#pragma unroll
        for (int k = 0; k < 8; ++k) if (key == k) pick = MODE != MODE_FAST ? kPairsExact[k] : (LAY == 1 ? kPairsFast1[k] : kPairsFast0[k]);
        stage = __builtin_amdgcn_readfirstlane(two ? pick : wave);
    }
    if (wave == 0) atomicMax(maxLenP, d.length);
    __syncthreads();
    if (stage == FINAL) { rowBase[lane] = d.outStart; rowCount[lane] = 0; }   // read by this wave only
    const uint32_t maxLen = *maxLenP;
    const int nChunks = (int)((maxLen + kChunk - 1) / kChunk);
    const int nIter = nChunks + 5;   // the final stage lags 5 chunks; same trip count in every wave
#define PIPE(p, c, i) (p)[(((c) & 1) * kChunk + (i)) * kLanes + lane]
    auto noChunk = [&](uint32_t) __attribute__((always_inline)) {};
    auto ctx = [&](auto stageTag) __attribute__((always_inline)) {
        constexpr int S = decltype(stageTag)::value, kFirst = direct_stage_first(S, LAY);
        return DirectCtx{A, A.directHdr + (size_t)S * A.nDirect, reinterpret_cast<const dpair*>(A.directRec) + (size_t)kFirst * A.nDirect};
    };

    if (stage == 0 && (ONLY & 1)) {
        // ================= T0: glottal source + aspiration noise (layout 1: pitch, vibrato, phase) =================
        using DD = DirectDesc<0, LAY>;
        const DirectCtx X = ctx(std::integral_constant<int, 0>{});
        direct_source_stage<DD, MODE, CH, LEAN, LAY == 1>(A, d, live, u, rec0, X, lane, nIter, nChunks, pipeX0, nkey, ninc, ninc2);
    } else if (LAY == 1 && stage == 1 && (ONLY & 64)) {
        // ================= layout 1, T1: glottal wave + aspiration noise from the phase (reference src/speechWaveGenerator.cpp:63-86) =================
        if constexpr (LAY == 1) {
        using DD = DirectDesc<1, LAY>; // gains: (turbulence, openQuotient) (voiceAmplitude, aspirationAmplitude) (preFormantGain, -)
        const DirectCtx X = ctx(std::integral_constant<int, 1>{});
        DirectState<DD, MODE, LEAN> f;
        direct_init(f, live, d, rec0, X);
        double aspNoise = 0.0;
        uint32_t noiseSt = noise_first(nkey, ninc);    // aspiration: noise values 0, 2, 4, ...
        direct_loop<DD, MODE, CH, LEAN>(1, nIter, nChunks, stage, f, X,
            [&](int c, int i) __attribute__((always_inline)) { return In1{PIPE(pipeX0, c, i)}; },
            [&](int c, int i, const In1& in, const auto& mid) __attribute__((always_inline)) {
                const double pitchPhase = in.a;
                const double turbGain = f.gv[0].x, openQ = f.gv[0].y, voiceAmp = f.gv[1].x, aspAmp = f.gv[1].y, preGain = f.gv[2].x;
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
                PIPE(pipeX1, c, i) = out;
                mid(out);
            },
            noChunk);
        }
    } else if (stage == (LAY == 1 ? 2 : 1) && (ONLY & 2)) {
        // ================= N0 (anti), NP mixed in by caNP (reference src/speechWaveGenerator.cpp:149-152) =================
        constexpr int ST = LAY == 1 ? 2 : 1;
        using DD = DirectDesc<ST, LAY>;      // gains: (caNP, -)
        const DirectCtx X = ctx(std::integral_constant<int, ST>{});
        double* const pin = LAY == 1 ? pipeX1 : pipeX0;
        double* const pout = LAY == 1 ? pipeX2 : pipeX1;
        DirectState<DD, MODE, LEAN> f;
        direct_init(f, live, d, rec0, X);
        direct_loop<DD, MODE, CH, LEAN>(ST, nIter, nChunks, stage, f, X,
            [&](int c, int i) __attribute__((always_inline)) { return In1{PIPE(pin, c, i)}; },
            [&](int c, int i, const In1& in, const auto& mid) __attribute__((always_inline)) {
                const double x = in.a;
                const double n0 = dot3<MODE>(f.a(0), x, f.b(0), f.z1[0], f.c(0), f.z2[0]);
                f.z2[0] = f.z1[0]; f.z1[0] = x;                              // the anti-resonator remembers its INPUT (reference :133)
                const double np = f.step(1, n0);
                const double o = fade_value(x, np, f.gv[0].x);
                PIPE(pout, c, i) = o;
                mid(o);      // the next sample's values: every coefficient and gain of this one has been used
            },
            noChunk);
    } else if (stage >= (LAY == 1 ? 3 : 2) && stage <= 4 && (ONLY & 4)) {
        // ================= the cascade: layout 0 T2, T3, T4 two resonators each (r6 r5 | r4 r3 | r2 r1); layout 1 T3, T4 three each =================
        // The cascade stages run the SAME program on different pipes and records: ONE copy of it, with the stage's pipes, records and depth
        // as run-time values (three instantiations were 25 KB of the MODE_EXACT kernel's ~69 KB of loops, all hot on a CU at once, against
        // a 64 KB instruction cache shared by two CUs).
        constexpr int ST = 4;                             // (any of them: same resonator and gain counts; the stage number only names the records)
        using DD = DirectDesc<ST, LAY>;
        static_assert(LAY == 1 || (direct_stage_res(2, LAY) == direct_stage_res(4, LAY) && direct_stage_gains(2, LAY) == direct_stage_gains(4, LAY)), "the cascade stages are alike");
        static_assert(direct_stage_res(3, LAY) == direct_stage_res(4, LAY) && direct_stage_gains(3, LAY) == direct_stage_gains(4, LAY), "the cascade stages are alike");
        double* const pin = stage == 2 ? pipeX1 : (stage == 3 ? pipeX2 : pipeX3);
        double* const pout = stage == 2 ? pipeX2 : (stage == 3 ? pipeX3 : pipeO);
        const int kFirst = stage == 2 ? direct_stage_first(2, LAY) : (stage == 3 ? direct_stage_first(3, LAY) : direct_stage_first(4, LAY));
        const DirectCtx X{A, A.directHdr + (size_t)stage * A.nDirect, reinterpret_cast<const dpair*>(A.directRec) + (size_t)kFirst * A.nDirect};
        DirectState<DD, MODE, LEAN> f;
        direct_init(f, live, d, rec0, X);
        direct_loop<DD, MODE, CH, LEAN>(stage, nIter, nChunks, stage, f, X,
            [&](int c, int i) __attribute__((always_inline)) { return In1{PIPE(pin, c, i)}; },
            [&](int c, int i, const In1& in, const auto& mid) __attribute__((always_inline)) {
                double o = in.a;
#pragma unroll
                for (int r = 0; r < DD::NRES; ++r) o = f.step(r, o);
                PIPE(pout, c, i) = o;
                mid(o);
            },
            noChunk);
    } else if (stage == 5 && (ONLY & 8)) {
        // ================= T5: frication noise, parallel 1, 2 =================
        using DD = DirectDesc<5, LAY>; // gains: (fricationAmplitude, preFormantGain) (pa1, pa2)
        const DirectCtx X = ctx(std::integral_constant<int, 5>{});
        DirectState<DD, MODE, LEAN> f;
        direct_init(f, live, d, rec0, X);
        double fricNoise = 0;
        uint32_t noiseSt = noise_step(noise_first(nkey, ninc), ninc);     // frication: noise values 1, 3, 5, ...
        direct_loop<DD, MODE, CH, LEAN>(3, nIter, nChunks, stage, f, X,
            [&](int, int) __attribute__((always_inline)) { return In0{}; },
            [&](int c, int i, const In0&, const auto& mid) __attribute__((always_inline)) {
                fricNoise = noise_uniform(noiseSt) + 0.75 * fricNoise;
                noiseSt = noise_step2(noiseSt, ninc2);
                const double fric = fricNoise * 0.3 * f.gv[0].x;
                const double y = (fric * f.gv[0].y) * 0.5;
                double par = 0;
                double w = f.step(0, y);
                par += (w - y) * f.gv[1].x;
                w = f.step(1, y);
                par += (w - y) * f.gv[1].y;
                PIPE(pipeY, c, i) = y; PIPE(pipeP, c, i) = par;
                mid(par);
            },
            noChunk);
    } else if (stage == 6 && (ONLY & 16)) {
        // ================= T6: parallel 3, 4 (the sum continues in the reference's order; y travels on) =================
        using DD = DirectDesc<6, LAY>; // gains: (pa3, pa4)
        const DirectCtx X = ctx(std::integral_constant<int, 6>{});
        DirectState<DD, MODE, LEAN> f;
        direct_init(f, live, d, rec0, X);
        direct_loop<DD, MODE, CH, LEAN>(4, nIter, nChunks, stage, f, X,
            [&](int c, int i) __attribute__((always_inline)) { return In2{PIPE(pipeY, c, i), PIPE(pipeP, c, i)}; },
            [&](int c, int i, const In2& in, const auto& mid) __attribute__((always_inline)) {
                const double y = in.a;
                double par = in.b;
                double w = f.step(0, y);
                par += (w - y) * f.gv[0].x;
                w = f.step(1, y);
                par += (w - y) * f.gv[0].y;
                PIPE(pipeY2, c, i) = y; PIPE(pipeP2, c, i) = par;
                mid(par);
            },
            noChunk);
    } else if (stage == FINAL && (ONLY & 32)) {
        // ================= T7: parallel 5, 6, bypass | cascade + parallel, gain, clip, int16 -> PCM =================
        using DD = DirectDesc<7, LAY>; // gains: (pa5, pa6) (parallelBypass, outputGain)
        const DirectCtx X = ctx(std::integral_constant<int, 7>{});
        DirectState<DD, MODE, LEAN> f;
        direct_init(f, live, d, rec0, X);
        int16_t* const myRow = reinterpret_cast<int16_t*>(tile + lane * kTileStride);
        uint32_t it = 0;
        // the tile leaves as 16 bytes per lane: lane -> (row, 8-sample piece), four pieces per 32-sample row, sixteen rows per pass.
        // PASSES rows' worth of LDS reads in flight at a time: all eight with 256 registers, two with the lean stages' 128.
        auto flush_tile = [&](uint32_t tileStart, uint32_t validTo, uint32_t produced) __attribute__((always_inline)) {
            rowCount[lane] = produced;
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // this wave alone owns the tile: wave-level ordering is enough
            constexpr int kChunksPerRow = kTile / 8;
            constexpr int kRowsPerPass = kLanes / kChunksPerRow;
            constexpr int kPasses = kLanes / kRowsPerPass;
            constexpr int kGroup = LEAN ? 2 : kPasses;
            int ln = lane;
            if (LEAN) asm volatile("" : "+v"(ln));      // (the lane's row and piece are derived HERE: hoisted out of the loops they are registers held, or spilled, for the whole launch)
            const int chunk = ln % kChunksPerRow;
            const uint32_t first = tileStart + (uint32_t)chunk * 8u;
#pragma unroll
            for (int p0 = 0; p0 < kPasses; p0 += kGroup) {
                uint2 lo[kGroup], hi[kGroup];
                uint32_t cnt[kGroup];
                long long base[kGroup];
#pragma unroll
                for (int p = 0; p < kGroup; ++p) {
                    const int row = (p0 + p) * kRowsPerPass + ln / kChunksPerRow;
                    const uint2* src = reinterpret_cast<const uint2*>(tile + row * kTileStride + chunk * 16);
                    lo[p] = src[0]; hi[p] = src[1];
                    cnt[p] = rowCount[row];
                    base[p] = rowBase[row];
                }
#pragma unroll
                for (int p = 0; p < kGroup; ++p) {
                    if (cnt[p] > first && first < validTo) {
                        uint4* dst = reinterpret_cast<uint4*>(A.pcm + base[p] + first);
                        *dst = make_uint4(lo[p].x, lo[p].y, hi[p].x, hi[p].y);
                    }
                }
                if (LEAN) asm volatile("" ::: "memory");      // (one group's loads and stores before the next group's: the compiler would batch all eight again)
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        };
        direct_loop<DD, MODE, CH, LEAN>(5, nIter, nChunks, stage, f, X,
            [&](int c, int i) __attribute__((always_inline)) { return In3{PIPE(pipeO, c, i), PIPE(pipeY2, c, i), PIPE(pipeP2, c, i)}; },
            [&](int c, int i, const In3& in, const auto& mid) __attribute__((always_inline)) {
                const double o = in.a;
                const double y = in.b;
                double par = in.c;
                double w = f.step(0, y);
                par += (w - y) * f.gv[0].x;
                w = f.step(1, y);
                par += (w - y) * f.gv[0].y;
                par = fade_value(par, y, f.gv[1].x);
                const double mix = o + par;
                const double v = (mix * f.gv[1].y) * 4000.0;
                const double lo = (v < 32000.0) ? v : 32000.0;       // windows.h min(): NaN -> 32000
                const double cl = (lo > -32000.0) ? lo : -32000.0;
                myRow[(it % kTile) + i] = (int16_t)(uint32_t)(int)cl;   // (int) truncates toward zero (reference :208)
                mid(cl);
            },
            [&](uint32_t end) __attribute__((always_inline)) { it += kChunk; if ((it % kTile) == 0) flush_tile(it - kTile, it, f.length < end ? f.length : end); });
        if ((it % kTile) != 0) flush_tile(it - (it % kTile), it, f.length);
    } else {
        // (register census builds, ONLY: a stage left out still takes part in every barrier)
        for (int iter = 0; iter < nIter; ++iter) __syncthreads();
    }
#undef PIPE
}

}  // namespace klatt
