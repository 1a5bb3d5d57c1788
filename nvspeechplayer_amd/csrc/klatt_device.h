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
// each lane does here for its own utterance:
//   * the "old" and "new" frame of the running fade are staged in LDS ([param][lane],
//     conflict-free ds_read_b64); the per-sample working set (filter memories,
//     coefficients, the parameters the DSP reads) lives in VGPRs;
//   * a request is E F..F E S..S: an event sample (dequeue), F fade samples, an event
//     sample (fade end), then steady samples.  The wavefront runs 8-sample blocks of
//     branch-free steady or fade code whenever every live lane is inside such a stretch,
//     and single general steps around events;
//   * resonator coefficients are recomputed only on fade samples that can move them;
//   * branches that provably contribute nothing (noise sources and the parallel bank of an
//     utterance whose frames have zero noise gains; vibrato at zero depth) are skipped;
//   * PCM goes to a padded LDS tile (ds_write_b16, conflict-free) and leaves as 16 B per
//     lane, whole 64-byte row segments per utterance.
//
// Arithmetic is IEEE double.  This translation unit is compiled with -ffp-contract=off.
// MODE_EXACT rounds every multiply and add of the signal path separately, as the reference binary does;
// MODE_FAST fuses the resonators' multiply-adds.  Both evaluate exp/cos with klatt_math.h (<= 1 ulp).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

#include "klatt_math.h"

namespace klatt {

constexpr int kNumParams = 47;
constexpr int kLanes = 64;
constexpr int kTile = 32;                 // samples per lane per output tile (64-byte row segments)
constexpr int kTileStride = kTile * 2 + 8; // bytes per tile row; the pad keeps ds_write_b16 conflict-free
constexpr int kSlots = 45;                // parameters 1..45 live in LDS slots 0..44
constexpr int kNumRes = 14;
constexpr int kBlock = 16;                // samples per specialised block (two per tile)
constexpr int kStateDoubles = 240;        // per-stream saved state (streaming path)

constexpr int MODE_EXACT = 0;
constexpr int MODE_FAST = 1;

constexpr uint32_t FRAME_NULL = 1u;       // FrameMeta.flags
constexpr uint32_t UTT_NEEDS_NOISE = 1u;  // UttDesc.flags
constexpr uint32_t UTT_TRACKED = 4u;      // UttDesc.flags: noisy, every parameter finite, tracks planned (2u: UTT_NO_NASAL, klatt_lanepipe.h)
constexpr uint32_t UTT_DIRECT = 8u;       // UttDesc.flags: noisy, every parameter finite and in the range of klatt_math.h, no tracks: direct stages (klatt_direct.h)
constexpr int kUttKindShift = 8;          // UttDesc.flags bits 8..31 of a tracked utterance: the entry kinds (klatt_tracks) whose values change after the first sample
                                          // of its first fade -- a kind outside the mask of every lane of a wavefront is loaded once and never again (flat stages)

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
    uint32_t flags;          // UTT_NEEDS_NOISE: some frame has a non-zero (or non-finite) noise gain
    uint32_t length;         // samples this utterance produces (closed form, host-computed)
};

struct UttResult {           // written by the kernel
    uint32_t produced;       // samples written by this launch
    uint32_t framesTaken;    // frames dequeued by this launch
    int32_t lastIndex;
    uint32_t drained;        // 1: the queue ran dry (short count)
};

// ---- tracks (klatt_tracks.h) ---------------------------------------------------------------------------------
// A resonator's coefficients are a pure function of (f, bw) (reference src/speechWaveGenerator.cpp:112-127), and during a
// fade every parameter is a pure function of the fade's two end frames and the sample counter (reference src/frame.cpp:48-53):
// what the filter stages need on a fade sample does not depend on the signal.  The tracked kernels therefore take it from a
// TRACK -- evaluated densely (lanes = entries) by klatt_tracks before the synthesis launch -- instead of interpolating and
// evaluating exp/cos inside the sample recurrence.  A track is an array of 16-byte entries, kTrackEntries kinds of them:
//   0..13   resonator r (N0, NP, c6..c1, p1..p6): (b, c); a = 1 - b - c, except for the anti-resonator N0, whose a is a
//           second entry (a, -) right after its (b, c)
//   14..19  the interpolated gains of the filter stages, in pairs: (caNP, -) | (pa5, pa6) (parallelBypass, outputGain) |
//           (fricationAmplitude, preFormantGain) (pa1, pa2) (pa3, pa4)          -- S1 | final stage | parallel stage
//   20..23  the source stage's parameters: (vibratoPitchOffset, vibratoSpeed) (voiceTurbulenceAmplitude, glottalOpenQuotient)
//           (voiceAmplitude, aspirationAmplitude) (preFormantGain, -)           -- S0 (the pitch itself glides: not in a track)
// On fade sample 1 everything is re-evaluated (see fade_update below): every kind has an entry for it, kTrackFirst entries (N0
// takes two).  Fade samples 2..F change only the kinds that MOVE in the fade (mask).  A fade's track is four PARTS, one per flat
// stage (S0: kinds 0, 1, 14, 20..23 | S1: 2..7 | final stage: 12, 13, 15, 16 | parallel stage: 8..11, 17..19 -- KLATT_FLAT_LAYOUT below), so that what a
// stage streams through is contiguous and no cache line is shared by two stages -- which work 16 to 32 samples apart: with the
// kinds of all stages interleaved in one row a line was fetched once per stage, 34 GB from HBM per launch of a batch with 445 MB
// of tracks, for 22 GB of rows.  Layout of a part:
//   header   the stage's kinds that do NOT move, ascending: their value of fade sample 1 (which is their value on every later sample)
//   matrix   F rows of the stage's kinds that move, ascending (N0 takes two); row j holds fade sample j + 1
// so a moving kind's entries are one stride (the part's moving entries) apart from the first fade sample on, and a kind that
// does not move has ONE entry: a stage points at it with stride 0.  kTrackFirst + (F - 1) * nSlots entries in all.
// Fades with bitwise equal end values of all the parameters involved and the same length share one track (host, plan_tracks).
constexpr int kTrackEntries = 24;
constexpr int kTrackFirst = kTrackEntries + 1;   // entries of a fade's first sample (N0 takes two)
constexpr int kShapeValues = 45;                 // a SHAPE: the parameter values a track depends on at one end of the fade
constexpr int kShapeStride = 46;
// where the values of a shape come from in a frame: (f, bw) of resonator r at 2r, 2r + 1; then the gains
__host__ __device__ constexpr int shape_param(int i)
{
    constexpr int kF[14] = {13, 14, 12, 11, 10, 9, 8, 7, 25, 26, 27, 28, 29, 30};
    constexpr int kB[14] = {21, 22, 20, 19, 18, 17, 16, 15, 31, 32, 33, 34, 35, 36};
    constexpr int kG[17] = {23, 41, 42, 43, 45, 24, 44, 37, 38, 39, 40, 1, 2, 3, 4, 5, 6};
    return i < 28 ? ((i & 1) ? kB[i >> 1] : kF[i >> 1]) : kG[i - 28];
}
constexpr int kShapePreGain = 28 + 6;            // preFormantGain (parameter 44): silence gates it (reference src/frame.cpp:61,66)
// the two shape values of gain entry e (14..19); -1: none
__host__ __device__ constexpr int entry_value(int e, int half)
{
    constexpr int kV[10][2] = {{28, -1}, {29, 30}, {31, 32}, {33, 34}, {35, 36}, {37, 38}, {39, 40}, {41, 42}, {43, 44}, {34, -1}};
    return kV[e - 14][half];
}
struct TrackRef {            // per frame on the host (plan_tracks); the stages read it as a FlatRef
    unsigned long long off;  // first entry of the fade's track
    uint32_t mask;           // entry kinds that move in the fade (bit e)
    uint32_t nSlots;         // entries per fade sample after the first: popcount(mask) + (mask & 1)
};
struct FlatRef {             // 16 B per frame: what a flat stage needs when the frame's fade starts, in one load (host: setUtterances)
    uint32_t off;            // first entry of the fade's track (tracks hold fewer than 2^32 entries: 64 GB)
    uint32_t mask;           // entry kinds that move in the fade
    uint32_t fadeSamples;
    uint32_t span;           // samples from this frame's dequeue to the next frame's: max(minSamples, fadeSamples + 1) + 1
};
struct SourceRef {           // 32 B per frame: what the flat source stage needs when the frame is dequeued, in two loads (host: setUtterances)
    double pitch;            // voicePitch (parameter 0)
    double pitchInc;         // (endVoicePitch - voicePitch) / minSamples     (reference src/frame.cpp:98; the host's division is the device's: IEEE)
    double invFade;          // 1 / fadeSamples
    int32_t userIndex;
    uint32_t flags;          // FRAME_NULL
};
struct TrackJob {            // one distinct track, read by klatt_tracks
    unsigned long long off;
    uint32_t fromShape, toShape;    // the fade's end points (index into TrackArgs.shapes)
    uint32_t fadeSamples;
    uint32_t mask;
};
__host__ __device__ inline uint32_t track_slots(uint32_t mask) { return (uint32_t)__builtin_popcount(mask) + (mask & 1u); }
constexpr int kTrackStages = 4;
// the kinds of a stage's part, and their entries (N0 takes two)
// Which stage runs what (the flat stages of klatt_systolic.h; a stage's part of a track holds its kinds):
//   1: S0 source | S1 N0, NP, r6..r3 | final r2, r1, parallel 5, 6, mix, PCM | S3 frication, parallel 1..4
//   2: S0 source, N0, NP | S1 r6..r1 | final parallel 5, 6, mix, PCM | S3 as before    (the final stage's mix / clip / PCM tail is as
//      long as four resonators, the source stage was the lightest by a third: profiles/r3_stage_balance.txt)
#ifndef KLATT_FLAT_LAYOUT
#define KLATT_FLAT_LAYOUT 2
#endif
__host__ __device__ constexpr uint32_t track_stage_kinds(int s)
{
    return KLATT_FLAT_LAYOUT == 2 ? (s == 0 ? 0xF04003u : s == 1 ? 0x0000FCu : s == 2 ? 0x01B000u : 0x0E0F00u)
                                  : (s == 0 ? 0xF00000u : s == 1 ? 0x00403Fu : s == 2 ? 0x01B0C0u : 0x0E0F00u);
}
__host__ __device__ constexpr uint32_t track_stage_entries(int s)
{
    return KLATT_FLAT_LAYOUT == 2 ? (s == 0 ? 8u : s == 1 ? 6u : s == 2 ? 4u : 7u) : (s == 0 ? 4u : s == 1 ? 8u : s == 2 ? 6u : 7u);
}
constexpr int kTrackAntiStage = KLATT_FLAT_LAYOUT == 2 ? 0 : 1;      // the stage of N0, whose kind takes two entries
// entries of stage s's part that move: per row of its matrix
__host__ __device__ inline uint32_t track_stage_slots(uint32_t mask, int s) { return (uint32_t)__builtin_popcount(mask & track_stage_kinds(s)) + (s == kTrackAntiStage ? (mask & 1u) : 0u); }
// first entry of stage s's part in a track of a fade of F samples
__host__ __device__ inline uint32_t track_part(uint32_t mask, uint32_t F, int s)
{
    uint32_t at = 0;
    for (int k = 0; k < s; ++k) at += track_stage_entries(k) + (F - 1u) * track_stage_slots(mask, k);
    return at;
}

// ---- direct stages (klatt_direct.h): what a fade needs, per frame and stage, evaluated by klatt_seeds before the launch --------
struct DirectHdr {           // 16 B per (frame, stage), loaded one fade ahead (stages: direct_stage_kind below)
    uint32_t fadeSamples;
    uint32_t span;           // samples from this frame's dequeue to the next frame's: max(minSamples, fadeSamples + 1) + 1
    uint32_t bits;           // of the stage's kinds e = 0, 1, ...: bit e -- the fade moves kind e; bit 8 + r -- it moves resonator r's bandwidth;
                             // bits 14 + 2r, 15 + 2r -- resonator r's arguments leave the unreduced range / the range of cosine quadrant -1 somewhere in the fade
    uint32_t pad;
};
struct DirectJob {           // host -> klatt_seeds: the end points of one frame's fade (reference src/frame.cpp:55-72, walked on the host)
    uint32_t frame;          // the frame (durations: meta[frame])
    uint32_t from, to;       // the frames whose values the fade starts from / ends on; kNoFrame: all zero (a fresh handle)
    uint32_t flags;          // bit 0: the start's preFormantGain is gated off, bit 1: the end's (silence, reference src/frame.cpp:61,66)
};
constexpr uint32_t kNoFrame = 0xFFFFFFFFu;
constexpr uint32_t kDirectBwShift = 8, kDirectClsShift = 14, kDirectAllBits = 0x3FFFFFFu;
// Which direct stage runs what (kinds as in the tracks above; resonator kinds first).  EIGHT stages, one wavefront each:
//   T0 source | T1 N0, NP (caNP mix) | T2 r6, r5 | T3 r4, r3 | T4 r2, r1 | T5 frication, parallel 1, 2 | T6 parallel 3, 4 |
//   T7 parallel 5, 6, bypass mix, gain, clip, PCM
// -- two resonators per stage: what a direct stage carries per resonator (coefficients, memories, the running fade's end points:
// ten doubles) does not fit the register budget of a wavefront four or six at a time (klatt_direct.h has the census).
// A stage's record of one frame: 4 entries (16 B) per resonator kind, then 3 per gain kind (layouts: klatt_direct.h).
constexpr int kDirectStages = 8;
// Two layouts (L): 0 as above -- what MODE_EXACT runs, whose stages are bound by the coefficient polynomials, two resonators each --,
// and 1 for MODE_FAST, whose recurrences leave the SOURCE stage the slowest by far (88 instructions per row against 31 for a pair of
// cascade resonators): there the source is two stages and the cascade's six resonators are two stages of three,
//   T0 pitch, vibrato, phase | T1 glottal wave, aspiration noise, gains | T2 N0, NP | T3 r6, r5, r4 | T4 r3, r2, r1 | T5, T6, T7 as in layout 0.
#ifndef KLATT_DIRECT_FAST_LAYOUT
#define KLATT_DIRECT_FAST_LAYOUT 1
#endif
__host__ __device__ constexpr int direct_layout(int mode) { return mode == MODE_FAST ? KLATT_DIRECT_FAST_LAYOUT : 0; }
__host__ __device__ constexpr int direct_stage_res(int s, int L = 0)
{
    return L == 1 ? (s < 2 ? 0 : (s == 3 || s == 4) ? 3 : 2) : (s == 0 ? 0 : 2);
}
__host__ __device__ constexpr int direct_stage_gains(int s, int L = 0)
{
    return L == 1 ? (s == 0 ? 1 : s == 1 ? 3 : s == 2 ? 1 : s == 5 ? 2 : s == 6 ? 1 : s == 7 ? 2 : 0)
                  : (s == 0 ? 4 : s == 1 ? 1 : s == 5 ? 2 : s == 6 ? 1 : s == 7 ? 2 : 0);
}
__host__ __device__ constexpr int direct_stage_entries(int s, int L = 0) { return 4 * direct_stage_res(s, L) + 3 * direct_stage_gains(s, L); }
__host__ __device__ constexpr int direct_stage_first(int s, int L = 0) { int at = 0; for (int k = 0; k < s; ++k) at += direct_stage_entries(k, L); return at; }      // (no recursion: a recursive device function is a real call)
constexpr int kDirectEntries = direct_stage_first(kDirectStages);      // 16-byte entries per frame over the stages: 86 (1376 B), in both layouts
static_assert(direct_stage_first(kDirectStages, 1) == kDirectEntries, "both layouts hold the same kinds");
__host__ __device__ constexpr int direct_stage_kind(int s, int e, int L = 0)
{
    constexpr int k0[kDirectStages][4] = {{20, 21, 22, 23}, {0, 1, 14, -1}, {2, 3, -1, -1}, {4, 5, -1, -1}, {6, 7, -1, -1}, {8, 9, 17, 18}, {10, 11, 19, -1}, {12, 13, 15, 16}};
    constexpr int k1[kDirectStages][4] = {{20, -1, -1, -1}, {21, 22, 23, -1}, {0, 1, 14, -1}, {2, 3, 4, -1}, {5, 6, 7, -1}, {8, 9, 17, 18}, {10, 11, 19, -1}, {12, 13, 15, 16}};
    return L == 1 ? k1[s][e] : k0[s][e];
}

struct KernelArgs {
    const DirectHdr* directHdr;  // [8][nDirect] direct launches only: stage s's headers at directHdr + s * nDirect
    const double2* directRec;    // stage s's records at directRec + direct_stage_first(s) * nDirect, direct_stage_entries(s) entries per frame
    const uint32_t* directFirst; // [nUtt] the record number of an utterance's first frame
    uint32_t nDirect;            // frames of the launch's direct utterances
    const FlatRef* flatRef;      // [nFrames] tracked launches only
    const SourceRef* sourceRef;  // [nFrames] tracked launches only
    uint32_t trackBytes;         // size of `track` (below 4 GB: the flat stages address it through a buffer descriptor with 32-bit offsets)
    const double2* track;        // the launch's tracks
    const double* frames;        // [nFrames][47]
    const FrameMeta* meta;       // [nFrames]
    const UttDesc* utt;          // [nUtt]
    const uint32_t* order;       // [nSlots] utterance per lane slot, 0xFFFFFFFF = empty
    int16_t* pcm;
    UttResult* result;           // [nUtt]
    double* state;               // [nUtt][kStateDoubles] or nullptr (fresh streams, nothing saved)
    double* const* statePtrs;    // streaming: per-utterance state blocks (one per live handle); overrides `state`
    const uint32_t* control;     // [nUtt] streaming only: bit0 = apply purge before synthesising
    long long nSlots;
    uint32_t maxSamples;         // per launch and utterance; 0xFFFFFFFF = until drained
    uint32_t ringMask;           // 0: an utterance's frames are frameStart, frameStart + 1, ...; live handles: its frames sit in a ring of
                                 // ringMask + 1 frames (a power of two, aligned to its size) and frameStart points at the oldest one
    int sampleRate;
    double sampleRateF;          // (double)sampleRate
    double invSampleRate;        // RN(1/sr)
    double negPiOverSr;          // -pi/sr     (reference src/speechWaveGenerator.cpp:116)
    double twoPiOverSr;          // (2*pi)/sr  (reference src/speechWaveGenerator.cpp:118)
    unsigned long long* debug;   // diagnostic builds only (KLATT_STAMPS): per-wave cycle sums; nullptr otherwise
};

// resonator r reads frequency parameter kResF[r] and bandwidth parameter kResB[r]
// order: N0(anti), NP, c6, c5, c4, c3, c2, c1, p1..p6  (reference :149-156, :173-178)
__device__ constexpr int kResF[kNumRes] = {13, 14, 12, 11, 10, 9, 8, 7, 25, 26, 27, 28, 29, 30};
__device__ constexpr int kResB[kNumRes] = {21, 22, 20, 19, 18, 17, 16, 15, 31, 32, 33, 34, 35, 36};
// parameters the per-sample DSP reads directly (everything except 0, the f/bw pairs and 46)
constexpr int kNumHot = 17;
__device__ constexpr int kHot[kNumHot] = {1, 2, 3, 4, 5, 6, 23, 24, 37, 38, 39, 40, 41, 42, 43, 44, 45};

// ---- arithmetic helpers -------------------------------------------------------------------

// Correctly rounded x / b from y = RN(1/b) (Markstein): 3 instructions instead of a division
// sequence.  tests/native/check_math.cpp checks the identity against the '/' operator.
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
// The engine's noise definition (restated by oracle/klatt_oracle.c).  The reference draws (double)rand()/RAND_MAX twice per
// sample from the process-global rand() (src/speechWaveGenerator.cpp:40, called at :75 and :205): not reproducible across
// handles.  Here every utterance has a stream of its own: a 32-bit linear congruential generator (Numerical Recipes'
// multiplier) whose start AND increment come from the utterance's seed,
//     s_0 = noise_key(seed),  c = noise_inc(seed) (odd),   s_(n+1) = 1664525 s_n + c  (mod 2^32),   value k = s_(k+1) >> 1     (0 .. 2^31-1, glibc's range)
// and sample n takes value 2n for the aspiration and value 2n+1 for the frication, the order of the reference's two calls.
// Generators with different odd increments run through different full-period sequences, so two utterances do not replay each
// other's noise at a lag (round 2's generator had ONE increment: every stream was a window of the same 2^32-cycle, and a launch of
// BASELINE configs[2] draws 2.7e9 values, most of that cycle).  One multiply per value: the counter hash of round 1 (two multiplies
// and four xor-shifts per value) was 35 of the ~200-250 issue cycles of the source and frication stages per sample; this is 11.
// The stages step their own sub-sequence two values at a time (noise_step2).  A live handle keeps the state of its next
// aspiration value (stream slot 221); the increment is recomputed from the handle's seed.
__device__ __forceinline__ uint32_t noise_key(uint32_t seed) { return mix32(seed ^ 0x9E3779B9u); }
__device__ __forceinline__ uint32_t noise_inc(uint32_t seed) { return (mix32(seed + 0x85EBCA6Bu) << 1) | 1u; }
constexpr uint32_t kNoiseA = 1664525u;
constexpr uint32_t kNoiseA2 = kNoiseA * kNoiseA;                                       // two steps in one (mod 2^32): s * A^2 + (A + 1) c
__device__ __forceinline__ uint32_t noise_inc2(uint32_t inc) { return (kNoiseA + 1u) * inc; }
__device__ __forceinline__ uint32_t noise_step(uint32_t s, uint32_t inc) { return s * kNoiseA + inc; }
__device__ __forceinline__ uint32_t noise_step2(uint32_t s, uint32_t inc2) { return s * kNoiseA2 + inc2; }
__device__ __forceinline__ uint32_t noise_first(uint32_t key, uint32_t inc) { return noise_step(key, inc); }     // the state of value 0
// (double)rand()/RAND_MAX with RAND_MAX = 2^31-1 (reference src/speechWaveGenerator.cpp:40) of the value in state s.  For the
// integers r < 2^31 the correctly rounded quotient is fma(r, yh, r * yl) with 1/(2^31-1) = yh + yl to double-double
// (yh = 0x1.00000002p-31, yl = 2^-93; r * yl is exact): two operations instead of div_by's three.
// tests/native/check_math.cpp compares it with '/' for ALL 2^31 values.
__device__ __forceinline__ double noise_uniform(uint32_t s)
{
    const double r = (double)(s >> 1);
    return __builtin_fma(r, 0x1.00000002p-31, r * 0x1p-93);
}

// a*x + b*y + c*z in the reference's order: ((a*x) + (b*y)) + (c*z)
template <int MODE>
__device__ __forceinline__ double dot3(double a, double x, double b, double y, double c, double z)
{
    if (MODE == MODE_FAST) return __builtin_fma(c, z, __builtin_fma(b, y, a * x));
    return a * x + b * y + c * z;
}

// The type of the signal in the flat filter stages (klatt_systolic.h): resonator memories and coefficients, gains, the pipes
// between the stages.  double in every shipped build; -DKLATT_SIGNAL_F32 builds the float experiment of DESIGN.md section 4 ("A float signal path")
// (tools/f32_probe.py: how fast, how far from the double PCM).  Frames, tracks, the source stage (pitch, phases) and the
// coefficient evaluation are double in both.
#ifdef KLATT_SIGNAL_F32
typedef float sig_t;
#else
typedef double sig_t;
#endif
template <int MODE>
__device__ __forceinline__ float dot3(float a, float x, float b, float y, float c, float z)
{
    return __builtin_fmaf(c, z, __builtin_fmaf(b, y, a * x));
}
__device__ __forceinline__ float fade_value(float from, float to, float ratio)
{
    float v = from + ((to - from) * ratio);
    return (to != to) ? from : v;
}

// reference src/speechWaveGenerator.cpp:112-127
struct Coef { double a, b, c; };
// `cls` (wave-uniform): what the caller already knows about the arguments of every active lane -- COEF_UNREDUCED: no
// range reduction needed (exp k = 0, cos n = 0); COEF_QUADRANT_M1: exp k = 0, cos n = -1 (F3 and up);
// COEF_UNKNOWN: decide here, per evaluation, with ballots.  A whole-fade caller classifies once from the fade's end
// points (fade_classes in klatt_systolic.h): the ballots and the VALU -> SALU -> branch round trips per evaluation are
// what the short paths cost most.
enum { COEF_UNREDUCED = 0, COEF_QUADRANT_M1 = 1, COEF_UNKNOWN = 2 };
#ifndef KLATT_COLD_CALL
#define KLATT_COLD_CALL 2
#endif
struct RadCos { double rad, cs; };
#if KLATT_COLD_CALL
// exp and cos OUTSIDE the validated range of klatt_math.h (|arg| > 700 / 1e4: no frame a speech front-end produces): the device
// library's, out of line.  Inlined, its two dozen polynomial and reduction constants are materialised in the kernel's prologue and
// kept in registers through the hot loops (a 64-bit literal is no VALU operand on gfx950; see profiles/r2_isa_coefficient_block.txt):
// with the call the noisy stage kernel is 137 KB instead of 262 KB and spills 96 instead of 160 bytes per lane; cfg2 15.05 -> 14.6 ms,
// rotated 58.5 -> 55 ms, cfg3 14.8 -> 14.3, cfg4 9.84 -> 9.47 (tools/ab_probe.py).  KLATT_COLD_CALL = 1 moves the range-reduced
// fast_exp / fast_cos out of line as well: better still on aligned batches (14.4 ms), worse where lanes fade apart (62 ms).
// Results come back by value (registers): no address of a caller's local is taken.  Only the kernels call it (one level deep); the
// lane kernel's out-of-line resonator_coefficients() keeps the library calls inline (NESTED = false) -- a noinline function calling
// this one from inside a divergent branch gave wrong coefficients now and then (tests: MODE_FAST, layout 0, NaN parameters).
__device__ __attribute__((noinline)) RadCos exp_cos_reduced(double ex, double th)
{
    RadCos o;
#if KLATT_COLD_CALL == 1
    if (__builtin_fabs(ex) <= 700.0 && __builtin_fabs(th) <= 1.0e4) { o.rad = fast_exp(ex); o.cs = fast_cos(th); }
    else
#endif
    { o.rad = exp(ex); o.cs = cos(th); }
    return o;
}
#endif
// exp(-pi bw / sr) and cos(2 pi (-f) / sr) of one resonator (the two transcendental halves of resonator_coefficients_inline)
template <bool NESTED = true>
__device__ __forceinline__ RadCos coefficient_parts(double f, double bw, double negPiOverSr, double twoPiOverSr, int cls = COEF_UNKNOWN)
{
    const double ex = negPiOverSr * bw;
    const double th = twoPiOverSr * -f;
    double rad, cs;
    // exp and cos: the straight-line versions of klatt_math.h (<= 1 ulp, like a libm); arguments outside their
    // validated range take the device library.  Same code in both arithmetic modes.
    // Wave-uniform short cut (klatt_math.h): when no active lane's argument needs a range reduction the kernels
    // alone return the same bits as fast_exp / fast_cos.  (Separate decisions for exp and cos, and a third
    // variant for the quadrant of f > 2756 Hz, were measured slower: more branches and spills than they save.)
    const bool eu = exp_is_unreduced(ex);
    if (cls == COEF_UNREDUCED) { rad = exp_unreduced(ex); cs = cos_unreduced(th); }
    else if (cls == COEF_QUADRANT_M1) { rad = exp_unreduced(ex); cs = cos_quadrant_m1(th); }
    else if (__all(eu && cos_is_unreduced(th))) { rad = exp_unreduced(ex); cs = cos_unreduced(th); }
    else if (__all(eu && cos_is_quadrant_m1(th))) { rad = exp_unreduced(ex); cs = cos_quadrant_m1(th); }   // F3 and up
#if KLATT_COLD_CALL == 1
    else if (NESTED) { const RadCos o = exp_cos_reduced(ex, th); rad = o.rad; cs = o.cs; }
#endif
    else if (__builtin_fabs(ex) <= 700.0 && __builtin_fabs(th) <= 1.0e4) { rad = fast_exp(ex); cs = fast_cos(th); }
#if KLATT_COLD_CALL == 2
    else if (NESTED) { const RadCos o = exp_cos_reduced(ex, th); rad = o.rad; cs = o.cs; }
#endif
    else { rad = exp(ex); cs = cos(th); }
    RadCos o; o.rad = rad; o.cs = cs;
    return o;
}
// the coefficients from the two parts (reference src/speechWaveGenerator.cpp:117-126)
__device__ __forceinline__ Coef coefficient_finish(double rad, double cs, bool anti, double f)
{
    double cc = -(rad * rad);
    double bb = rad * cs * 2.0;
    double aa = 1.0 - bb - cc;
    if (anti && f != 0) {
        aa = 1.0 / aa;
        cc *= -aa;
        bb *= -aa;
    }
    Coef k; k.a = aa; k.b = bb; k.c = cc;
    return k;
}
template <int MODE, bool NESTED = true>
__device__ __forceinline__ Coef resonator_coefficients_inline(double f, double bw, bool anti, double negPiOverSr, double twoPiOverSr,
                                                              int cls = COEF_UNKNOWN)
{
    const RadCos p = coefficient_parts<NESTED>(f, bw, negPiOverSr, twoPiOverSr, cls);
    return coefficient_finish(p.rad, p.cs, anti, f);
}
// out-of-line copy for the lane kernel, whose 14 call sites would otherwise each inline exp and cos
template <int MODE>
__device__ __attribute__((noinline)) Coef resonator_coefficients(double f, double bw, bool anti, double negPiOverSr, double twoPiOverSr)
{
    return resonator_coefficients_inline<MODE, false>(f, bw, anti, negPiOverSr, twoPiOverSr);
}

// LDS per workgroup (one wavefront):
//   oldP[kSlots][64] f64, newP[kSlots][64] f64          46,080 B
//   curFB[28][64] f64 (STREAM only: current f/bw)        14,336 B
//   tile[64 rows][kTileStride B]                           4,608 B
//   rowBase[64] i64, rowCount[64] u32                        768 B
template <bool STREAM>
struct LdsLayout {
    static constexpr int kOld = 0;
    static constexpr int kNew = kOld + kSlots * kLanes * 8;
    static constexpr int kCurFB = kNew + kSlots * kLanes * 8;
    static constexpr int kTileOff = kCurFB + (STREAM ? 28 * kLanes * 8 : 0);
    static constexpr int kRowBase = kTileOff + kLanes * kTileStride;
    static constexpr int kRowCount = kRowBase + kLanes * 8;
    static constexpr int kBytes = kRowCount + kLanes * 4;
};

// ---- per-lane state ------------------------------------------------------------------------
struct Lane {
    double cur[kNumParams];    // only cur[0] and the kHot entries are ever touched (registers)
    double old0, new0, oldInc, newInc, invFade;
    double ra[kNumRes], rb[kNumRes], rc[kNumRes], z1[kNumRes], z2[kNumRes];
    double pitchPhase, vibPhase, aspNoise, fricNoise;
    uint32_t cnt, oldMin, newMin, newFade;
    uint32_t resMask, nextFrame, noiseState, produced;   // noiseState: the noise stream's state of the next aspiration value
    int32_t lastIndex;
    bool hasNew, oldNull, newNull, done, drained;
    bool vibFrames;            // the old or new frame has a non-zero (or NaN) vibrato depth/speed
};

// ---- one output sample from the lane's current parameters (reference :72-86, :147-180, :203-208)
// NOISE: the wavefront holds an utterance with non-zero noise gains; waveVib: some lane may have vibrato.
template <int MODE, bool NOISE>
__device__ __forceinline__ uint32_t dsp_sample(Lane& s, const KernelArgs& A, uint32_t ninc, bool waveVib)
{
    double vib = 1.0;
    if (waveVib) {   // wave-uniform
        // vibrato phase advances even at zero depth; fmod(0 + p, 1) == p so zero speed is a no-op
        const double vs = s.cur[2];
        const double adv = frac_toward_zero(div_by(vs, A.sampleRateF, A.invSampleRate) + s.vibPhase);
        s.vibPhase = (vs != 0.0) ? adv : s.vibPhase;
        vib = (sin(s.vibPhase * 6.283185307179586) * 0.06 * s.cur[1]) + 1.0;
    }
    s.pitchPhase = frac_toward_zero(div_by(s.cur[0] * vib, A.sampleRateF, A.invSampleRate) + s.pitchPhase);
    double voice = (s.pitchPhase * 2.0) - 1.0;
    double src;
    if (NOISE) {
        s.aspNoise = noise_uniform(s.noiseState) + 0.75 * s.aspNoise;     // :40
        double asp = s.aspNoise * 0.2;
        double turb = asp * s.cur[3];
        turb = (s.pitchPhase >= s.cur[4]) ? turb : turb * 0.01;               // glottis closed
        voice += turb;
        voice *= s.cur[5];
        asp *= s.cur[6];
        src = asp + voice;
    } else {
        // all noise gains are zero: turbulence and aspiration terms are exactly +0
        src = voice * s.cur[5];
    }

    // cascade (:147-158): N0 anti-resonator (memory takes the INPUT, :133), NP mixed in by caNP, r6..r1
    const double x = (src * s.cur[44]) * 0.5;
    double o;
    {
        const double n0 = dot3<MODE>(s.ra[0], x, s.rb[0], s.z1[0], s.rc[0], s.z2[0]);
        s.z2[0] = s.z1[0]; s.z1[0] = x;
        const double np = dot3<MODE>(s.ra[1], n0, s.rb[1], s.z1[1], s.rc[1], s.z2[1]);
        s.z2[1] = s.z1[1]; s.z1[1] = np;
        o = fade_value(x, np, s.cur[23]);
    }
#pragma unroll
    for (int r = 2; r < 8; ++r) {
        const double y = dot3<MODE>(s.ra[r], o, s.rb[r], s.z1[r], s.rc[r], s.z2[r]);
        s.z2[r] = s.z1[r]; s.z1[r] = y;
        o = y;
    }

    double mix = o;
    if (NOISE) {
        // frication + parallel bank (:205-206, :170-180)
        s.fricNoise = noise_uniform(noise_step(s.noiseState, ninc)) + 0.75 * s.fricNoise;
        s.noiseState = noise_step2(s.noiseState, noise_inc2(ninc));
        const double fric = s.fricNoise * 0.3 * s.cur[24];
        const double y = (fric * s.cur[44]) * 0.5;
        double par = 0.0;
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            const int r = 8 + k;
            const double w = dot3<MODE>(s.ra[r], y, s.rb[r], s.z1[r], s.rc[r], s.z2[r]);
            s.z2[r] = s.z1[r]; s.z1[r] = w;
            par += (w - y) * s.cur[37 + k];
        }
        par = fade_value(par, y, s.cur[43]);
        mix = o + par;
    }
    // else: frication gain 0 => the bank's input, memories and output stay exactly 0

    const double v = (mix * s.cur[45]) * 4000.0;
    const double lo = (v < 32000.0) ? v : 32000.0;       // windows.h min(): NaN -> 32000
    const double cl = (lo > -32000.0) ? lo : -32000.0;
    return (uint32_t)(int)cl;                             // (int) truncates toward zero (:208)
}

// ---- one fade sample's parameter update (reference src/frame.cpp:48-53) ----------------------
template <int MODE, bool STREAM, int NRES>
__device__ __forceinline__ void fade_update(Lane& s, const KernelArgs& A, const double* oldP, const double* newP,
                                            double* curFB, int lane)
{
    // ratio = (double)counter / numFadeSamples, correctly rounded
    const double ratio = div_by((double)s.cnt, (double)s.newFade, s.invFade);
    s.cur[0] = fade_value(s.old0, s.new0, ratio);
#pragma unroll
    for (int h = 0; h < kNumHot; ++h) {
        const int sl = kHot[h] - 1;
        s.cur[kHot[h]] = fade_value(oldP[sl * kLanes + lane], newP[sl * kLanes + lane], ratio);
    }
    // Coefficients are a pure function of (f, bw) (reference :112-127), so recomputing them whenever
    // the pair MAY have moved equals the reference's recompute-on-change: on the first fade sample
    // (the previous fade's last interpolated value need not equal the frame value), and afterwards
    // for resonators whose old and new (f, bw) differ.
    const uint32_t need = (s.cnt == 1) ? 0x3FFFu : s.resMask;
#pragma unroll
    for (int r = 0; r < NRES; ++r) {      // a quiet launch never runs the parallel bank (r >= 8)
        if (need & (1u << r)) {
            const int sf = kResF[r] - 1, sb = kResB[r] - 1;
            const double f = fade_value(oldP[sf * kLanes + lane], newP[sf * kLanes + lane], ratio);
            const double bw = fade_value(oldP[sb * kLanes + lane], newP[sb * kLanes + lane], ratio);
            if (STREAM) { curFB[(2 * r) * kLanes + lane] = f; curFB[(2 * r + 1) * kLanes + lane] = bw; }
            const Coef k = resonator_coefficients<MODE>(f, bw, r == 0, A.negPiOverSr, A.twoPiOverSr);
            s.ra[r] = k.a; s.rb[r] = k.b; s.rc[r] = k.c;
        }
    }
}

// may the lane's vibrato path be live?  (frames' parameters 1 and 2; NaN compares as non-zero)
__device__ __forceinline__ bool vib_in_frames(const double* oldP, const double* newP, int lane)
{
    return oldP[0 * kLanes + lane] != 0.0 || oldP[1 * kLanes + lane] != 0.0 ||
           newP[0 * kLanes + lane] != 0.0 || newP[1 * kLanes + lane] != 0.0;
}
__device__ __forceinline__ bool vib_live(const Lane& s)
{
    return s.vibFrames || s.cur[1] != 0.0 || s.cur[2] != 0.0 || s.vibPhase != s.vibPhase;
}

// Where an utterance's frames are: frame k of the queue is frames[base + ((off + k) & mask)] (KernelArgs.ringMask).
struct FrameWindow { long long base; uint32_t off, mask; };
__device__ __forceinline__ FrameWindow frame_window(const KernelArgs& A, const UttDesc& d)
{
    if (A.ringMask == 0) return FrameWindow{d.frameStart, 0u, 0xFFFFFFFFu};
    return FrameWindow{d.frameStart & ~(long long)A.ringMask, (uint32_t)d.frameStart & A.ringMask, A.ringMask};
}

// ---- an event sample: fade end, dequeue, or end of queue (reference src/frame.cpp:44-47, :54-75)
// Called with counter already incremented.  Returns true when a sample is emitted.
__device__ __forceinline__ bool event_step(Lane& s, const UttDesc& d, const double* myFrames, const FrameMeta* myMeta, uint32_t ringOff, uint32_t ringMask,
                                           double* oldP, double* newP, int lane)
{
    if (s.hasNew) {
        // fade finished: the new request becomes the old one (:44-47)
#pragma unroll
        for (int k = 0; k < kSlots; ++k) oldP[k * kLanes + lane] = newP[k * kLanes + lane];
        s.old0 = s.new0; s.oldMin = s.newMin; s.oldInc = s.newInc; s.oldNull = s.newNull;
        s.hasNew = false;
        return true;
    }
    if (s.nextFrame >= d.nFrames) {
        // queue empty: no current frame, generate() returns early (:74, wavegen :209-211)
        s.done = true;
        s.drained = true;
        return false;
    }
    // dequeue (:55-72)
    const uint32_t at = (ringOff + s.nextFrame) & ringMask;
    const FrameMeta m = myMeta[at];
    const double* g = myFrames + (size_t)at * kNumParams;
    s.nextFrame++;
    s.newMin = m.minSamples; s.newFade = m.fadeSamples; s.newNull = (m.flags & FRAME_NULL) != 0;
    if (s.newNull) {
        // silence keeps the old shape with the gain gated off (:59-63)
#pragma unroll
        for (int k = 0; k < kSlots; ++k) newP[k * kLanes + lane] = oldP[k * kLanes + lane];
        newP[(44 - 1) * kLanes + lane] = 0.0;
        s.new0 = s.cur[0];
        s.newInc = 0.0;
        s.resMask = 0;
    } else {
        const double g0 = g[0];
        const double g46 = g[46];
#pragma unroll
        for (int k = 0; k < kSlots; ++k) newP[k * kLanes + lane] = g[k + 1];
        s.new0 = g0;
        s.newInc = (g46 - g0) / (double)s.newMin;   // reference src/frame.cpp:98
        if (s.oldNull) {
            // coming out of silence: start from the new shape, gain 0 (:64-67)
#pragma unroll
            for (int k = 0; k < kSlots; ++k) oldP[k * kLanes + lane] = g[k + 1];
            oldP[(44 - 1) * kLanes + lane] = 0.0;
            s.old0 = g0;
            s.resMask = 0;
        } else {
            uint32_t mk = 0;
#pragma unroll
            for (int r = 0; r < kNumRes; ++r) {
                const double of = oldP[(kResF[r] - 1) * kLanes + lane];
                const double ob = oldP[(kResB[r] - 1) * kLanes + lane];
                const bool same = (g[kResF[r]] == of) && (g[kResB[r]] == ob);
                mk |= same ? 0u : (1u << r);
            }
            s.resMask = mk;
        }
    }
    if (m.userIndex != -1) s.lastIndex = m.userIndex;     // :69
    s.cnt = 0;                                            // :70
    s.new0 += s.newInc * (double)s.newFade;               // :71
    s.invFade = 1.0 / (double)s.newFade;
    s.hasNew = true;
    s.vibFrames = vib_in_frames(oldP, newP, lane);
    return true;
}

// ---- the kernel ---------------------------------------------------------------------------
// NOISE: the launch's utterances use their noise sources and parallel bank (host splits a batch into
// a quiet and a noisy group of wavefronts and launches each with its own instantiation).
template <int MODE, bool STREAM, bool NOISE>
__global__ void __launch_bounds__(kLanes) klatt_synthesize(const KernelArgs A)
{
    using L = LdsLayout<STREAM>;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    double* const oldP = reinterpret_cast<double*>(lds + L::kOld);
    double* const newP = reinterpret_cast<double*>(lds + L::kNew);
    double* const curFB = reinterpret_cast<double*>(lds + L::kCurFB);
    unsigned char* const tile = lds + L::kTileOff;
    long long* const rowBase = reinterpret_cast<long long*>(lds + L::kRowBase);
    uint32_t* const rowCount = reinterpret_cast<uint32_t*>(lds + L::kRowCount);

    const int lane = threadIdx.x;
    const long long slot = (long long)blockIdx.x * kLanes + lane;
    const uint32_t u = (slot < A.nSlots) ? A.order[slot] : 0xFFFFFFFFu;
    const bool live = (u != 0xFFFFFFFFu);

    UttDesc d;
    d.frameStart = 0; d.outStart = 0; d.nFrames = 0; d.seed = 0; d.flags = 0; d.length = 0;
    if (live) d = A.utt[u];
    const FrameWindow w = frame_window(A, d);
    const double* const myFrames = A.frames + w.base * kNumParams;
    const FrameMeta* const myMeta = A.meta + w.base;
    const uint32_t nkey = noise_key(d.seed), ninc = noise_inc(d.seed);

    // ---- fresh-handle state (reference src/frame.cpp:85-88, src/speechWaveGenerator.cpp:37,52,108-109)
    Lane s;
#pragma unroll
    for (int i = 0; i < kNumParams; ++i) s.cur[i] = 0.0;
    s.old0 = 0.0; s.new0 = 0.0; s.oldInc = 0.0; s.newInc = 0.0; s.invFade = 1.0;
    s.cnt = 0; s.oldMin = 0; s.newMin = 0; s.newFade = 1;
    s.hasNew = false; s.oldNull = true; s.newNull = false;
    s.lastIndex = -1; s.resMask = 0; s.nextFrame = 0; s.noiseState = noise_first(nkey, ninc); s.produced = 0;
    s.done = !live; s.drained = false; s.vibFrames = false;
#pragma unroll
    for (int r = 0; r < kNumRes; ++r) { s.ra[r] = 0.0; s.rb[r] = 2.0; s.rc[r] = -1.0; s.z1[r] = 0.0; s.z2[r] = 0.0; }
    s.pitchPhase = 0.0; s.vibPhase = 0.0; s.aspNoise = 0.0; s.fricNoise = 0.0;

#pragma unroll
    for (int k = 0; k < kSlots; ++k) { oldP[k * kLanes + lane] = 0.0; newP[k * kLanes + lane] = 0.0; }
    if (STREAM) {
#pragma unroll
        for (int k = 0; k < 28; ++k) curFB[k * kLanes + lane] = 0.0;
    }

    if (STREAM && live && (A.state || A.statePtrs)) {
        // ---- resume a stream (layout: see the save block at the end) ----
        const double* S = A.statePtrs ? A.statePtrs[u] : A.state + (size_t)u * kStateDoubles;
        if (S[239] != 0.0) {
#pragma unroll
            for (int k = 0; k < kSlots; ++k) { oldP[k * kLanes + lane] = S[k]; newP[k * kLanes + lane] = S[45 + k]; }
#pragma unroll
            for (int k = 0; k < 28; ++k) curFB[k * kLanes + lane] = S[90 + k];
#pragma unroll
            for (int h = 0; h < kNumHot; ++h) s.cur[kHot[h]] = S[118 + h];
            s.old0 = S[135]; s.new0 = S[136]; s.cur[0] = S[137];
#pragma unroll
            for (int r = 0; r < kNumRes; ++r) {
                s.ra[r] = S[138 + r]; s.rb[r] = S[152 + r]; s.rc[r] = S[166 + r]; s.z1[r] = S[180 + r]; s.z2[r] = S[194 + r];
            }
            s.pitchPhase = S[208]; s.vibPhase = S[209]; s.aspNoise = S[210]; s.fricNoise = S[211];
            s.oldInc = S[212]; s.newInc = S[213]; s.invFade = S[214];
            s.cnt = (uint32_t)S[215]; s.oldMin = (uint32_t)S[216]; s.newMin = (uint32_t)S[217]; s.newFade = (uint32_t)S[218];
            const uint32_t fl = (uint32_t)S[219];
            s.hasNew = fl & 1; s.oldNull = fl & 2; s.newNull = fl & 4;
            s.lastIndex = (int32_t)S[220]; s.noiseState = (uint32_t)S[221];
            // which resonators move in the running fade follows from the fade's end points (as event_step derived it); the
            // state may have been saved by the stage-parallel stream kernel, whose stages keep their own masks
            uint32_t mk = 0;
#pragma unroll
            for (int r = 0; r < kNumRes; ++r) {
                const bool same = (S[45 + kResF[r] - 1] == S[kResF[r] - 1]) && (S[45 + kResB[r] - 1] == S[kResB[r] - 1]);
                mk |= same ? 0u : (1u << r);
            }
            s.resMask = mk;
        }
        if (A.control && (A.control[u] & 1u)) {
            // purge (reference src/frame.cpp:103-112): cut over from the current interpolated frame
            s.cnt = s.oldMin;
            if (s.hasNew) {
                s.oldNull = s.newNull;
                s.old0 = s.cur[0];
#pragma unroll
                for (int h = 0; h < kNumHot; ++h) oldP[(kHot[h] - 1) * kLanes + lane] = s.cur[kHot[h]];
#pragma unroll
                for (int r = 0; r < kNumRes; ++r) {
                    oldP[(kResF[r] - 1) * kLanes + lane] = curFB[(2 * r) * kLanes + lane];
                    oldP[(kResB[r] - 1) * kLanes + lane] = curFB[(2 * r + 1) * kLanes + lane];
                }
                s.hasNew = false;
            }
        }
    }

    if (STREAM) s.vibFrames = vib_in_frames(oldP, newP, lane);

    rowBase[lane] = d.outStart;
    rowCount[lane] = 0;
    __syncthreads();

    int16_t* const myRow = reinterpret_cast<int16_t*>(tile + lane * kTileStride);
    uint32_t it = 0;    // wave-uniform: samples stepped by this launch (position in the output rows)

    // write out tile rows: 16 B per lane; lane -> (row, chunk), kTile/8 chunks per row
    auto flush_tile = [&](uint32_t tileStart, uint32_t validTo) __attribute__((always_inline)) {
        rowCount[lane] = s.produced;
        __syncthreads();
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
        __syncthreads();
    };

    // one general step: per lane an event, fade or steady sample
    auto general_step = [&]() __attribute__((always_inline)) {
        bool emit = false;
        if (!s.done) {
            s.cnt++;
            const bool fading = s.hasNew && s.cnt <= s.newFade;
            const bool steady = !s.hasNew && s.cnt <= s.oldMin;
            if (fading) {
                fade_update<MODE, STREAM, (NOISE ? kNumRes : 8)>(s, A, oldP, newP, curFB, lane);
                emit = true;
            } else if (steady) {
                s.cur[0] += s.oldInc;          // glide the pitch (reference src/frame.cpp:76-79)
                s.old0 = s.cur[0];
                emit = true;
            } else {
                emit = event_step(s, d, myFrames, myMeta, w.off, w.mask, oldP, newP, lane);
            }
        }
        const bool waveVib = __any(emit && vib_live(s));
        if (emit) {
            myRow[it % kTile] = (int16_t)dsp_sample<MODE, NOISE>(s, A, ninc, waveVib);
            s.produced++;
        }
        it++;
    };

    // kBlock branch-free samples; KIND 0 = every live lane steady, 1 = every live lane fading
    auto block_run = [&](auto kindTag) __attribute__((always_inline)) {
        constexpr int KIND = decltype(kindTag)::value;
        const uint32_t tpos = it % kTile;
        if (!s.done) {
#pragma nounroll
            for (int i = 0; i < kBlock; ++i) {
                s.cnt++;
                if (KIND == 0) { s.cur[0] += s.oldInc; s.old0 = s.cur[0]; }
                else fade_update<MODE, STREAM, (NOISE ? kNumRes : 8)>(s, A, oldP, newP, curFB, lane);
                myRow[tpos + i] = (int16_t)dsp_sample<MODE, NOISE>(s, A, ninc, false);
            }
            s.produced += kBlock;
        }
        it += kBlock;
    };

    {
        while (true) {
            if (STREAM && !s.done && s.produced >= A.maxSamples) s.done = true;
            if (!__any(!s.done)) break;
            // samples left in the lane's current stretch (0 = the next sample is an event)
            uint32_t rem = s.hasNew ? (s.newFade - s.cnt) : (s.oldMin > s.cnt ? s.oldMin - s.cnt : 0u);
            if (STREAM) rem = min(rem, A.maxSamples - s.produced);
            const bool roomy = s.done || rem >= (uint32_t)kBlock;
            const bool fits = (it % kTile) + kBlock <= (uint32_t)kTile;
            int kind = -1;
            if (fits && __all(roomy) && !__any(!s.done && vib_live(s))) {
                if (!__any(!s.done && s.hasNew)) kind = 0;
                else if (!__any(!s.done && !s.hasNew)) kind = 1;
            }
            if (kind == 0) block_run(std::integral_constant<int, 0>());
            else if (kind == 1) block_run(std::integral_constant<int, 1>());
            else general_step();
            if ((it % kTile) == 0) flush_tile(it - kTile, it);
        }
    }

    if ((it % kTile) != 0) flush_tile(it - (it % kTile), it);

    if (live) {
        UttResult res;
        res.produced = s.produced; res.framesTaken = s.nextFrame; res.lastIndex = s.lastIndex; res.drained = s.drained ? 1u : 0u;
        A.result[u] = res;
    }

    if (STREAM && live && (A.state || A.statePtrs)) {
        double* S = A.statePtrs ? A.statePtrs[u] : A.state + (size_t)u * kStateDoubles;
#pragma unroll
        for (int k = 0; k < kSlots; ++k) { S[k] = oldP[k * kLanes + lane]; S[45 + k] = newP[k * kLanes + lane]; }
#pragma unroll
        for (int k = 0; k < 28; ++k) S[90 + k] = curFB[k * kLanes + lane];
#pragma unroll
        for (int h = 0; h < kNumHot; ++h) S[118 + h] = s.cur[kHot[h]];
        S[135] = s.old0; S[136] = s.new0; S[137] = s.cur[0];
#pragma unroll
        for (int r = 0; r < kNumRes; ++r) {
            S[138 + r] = s.ra[r]; S[152 + r] = s.rb[r]; S[166 + r] = s.rc[r]; S[180 + r] = s.z1[r]; S[194 + r] = s.z2[r];
        }
        S[208] = s.pitchPhase; S[209] = s.vibPhase; S[210] = s.aspNoise; S[211] = s.fricNoise;
        S[212] = s.oldInc; S[213] = s.newInc; S[214] = s.invFade;
        S[215] = (double)s.cnt; S[216] = (double)s.oldMin; S[217] = (double)s.newMin; S[218] = (double)s.newFade;
        S[219] = (double)((s.hasNew ? 1u : 0u) | (s.oldNull ? 2u : 0u) | (s.newNull ? 4u : 0u));
        S[220] = (double)s.lastIndex; S[221] = (double)s.noiseState;
        S[239] = 1.0;
    }
}

}  // namespace klatt
