// klatt_tracks.h -- tracks: what the synthesis needs on every fade sample, evaluated densely beforehand.
//
// The reference recomputes a resonator's coefficients whenever its frequency or bandwidth changed (reference
// src/speechWaveGenerator.cpp:112-127), i.e. on every sample of a fade for every resonator the fade moves, and interpolates every
// parameter on every sample of a fade (reference src/frame.cpp:48-53): inside the sample recurrence.  Inside the synthesis kernels
// that work runs with lanes = utterances, so a wavefront pays for it on every sample in which ANY of its 64 utterances is
// fading -- always, once the utterances of a wavefront are not copies of one sentence -- and its code (exp, cos, their polynomial
// constants, the interpolation, the frame state machine) shares the register budget of the filter stages.
//
// But the values do not depend on the signal: a parameter on fade sample n is
//     from + ((to - from) * (n / fadeSamples))                       (reference src/utils.h:20-23)
// of the fade's two end points.  This kernel evaluates them with lanes = ENTRIES of a track (a fade's samples x the entry kinds
// it moves; layout in klatt_device.h), every lane busy whatever the utterances' alignment, with the very functions the
// untracked stages call (fade_value, resonator_coefficients_inline: same operations, operands and rounding, so the tracked
// kernel produces the same PCM bit for bit).  The host gives equal fades ONE track (plan_tracks: a voice has a few dozen
// phoneme targets, so the fades of a batch are few distinct transitions, whatever the text), which makes the tracks a small,
// cache-resident table and this launch a few microseconds.  The flat stages of klatt_systolic.h pick the entries up.
#pragma once

#include "klatt_device.h"

namespace klatt {

struct TrackArgs {
    const TrackJob* jobs;
    long long nJobs;
    const double* shapes;        // [nShapes][kShapeStride]: the parameter values at the end points of the fades (host, plan_tracks)
    double2* track;
    double negPiOverSr, twoPiOverSr;
};

constexpr int kTrackWaves = 16;  // wavefronts per workgroup; a workgroup evaluates ONE track, wave w the entries [64 w, 64 w + 64), + 1024, ...
                                 // (BASELINE configs[2] has 94 tracks of a few thousand entries: one wave per track took 59 us)

__global__ void __launch_bounds__(kLanes * kTrackWaves) klatt_tracks(const TrackArgs T)
{
    const int lane = threadIdx.x & (kLanes - 1);
    const long long j = blockIdx.x;
    const TrackJob job = T.jobs[j];
    const uint32_t mask = job.mask;
    const double nf = (double)job.fadeSamples, invFade = 1.0 / nf;
    const double* const fo = T.shapes + (size_t)job.fromShape * kShapeStride;
    const double* const fn = T.shapes + (size_t)job.toShape * kShapeStride;
    double2* const out = T.track + job.off;
    // part by part (klatt_device.h: S0 | S1 | final stage | parallel stage), each its header, then its matrix
    uint32_t partAt = 0;
    for (int st = 0; st < kTrackStages; ++st) {
    const uint32_t kinds = track_stage_kinds(st), nS = track_stage_slots(mask, st), div = nS ? nS : 1u;
    const uint32_t header = track_stage_entries(st) - nS, total = track_stage_entries(st) + (job.fadeSamples - 1u) * nS;
    for (uint32_t e0 = (threadIdx.x >> 6) * kLanes; e0 < total; e0 += kLanes * kTrackWaves) {
        // every lane evaluates (the last pass repeats the part's last entry in its idle lanes): the wave-uniform short cuts of
        // resonator_coefficients_inline ballot over a full wavefront
        const uint32_t e = min(e0 + (uint32_t)lane, total - 1u);
        uint32_t cnt, r = 0u;     // r: the entry's kind
        bool second = false;      // N0's second entry: its a
        {
            // header (the kinds that do not move, fade sample 1), then the matrix (row n: fade sample n + 1 of the kinds that move)
            const bool hdr = e < header;
            const uint32_t q = hdr ? e : e - header, n = hdr ? 0u : q / div, s = hdr ? q : q - n * div;
            cnt = 1u + n;
            uint32_t accM = 0, accH = 0;
            for (int k = 0; k < kTrackEntries; ++k) {
                if (!((kinds >> k) & 1u)) continue;      // wave-uniform
                const bool moves = (mask >> k) & 1u;     // wave-uniform
                const uint32_t w = k == 0 ? 2u : 1u;
                uint32_t& acc = moves ? accM : accH;
                if (moves != hdr && s >= acc && s < acc + w) { r = (uint32_t)k; second = (k == 0 && s == acc + 1u); }
                acc += w;
            }
        }
        const double ratio = div_by((double)cnt, nf, invFade);
        // a resonator's (f, bw), or a gain entry's two values (a resonator's path is taken by the gain lanes too, on the values
        // of resonator 0, so that the wave-uniform short cuts of resonator_coefficients_inline see a full wavefront)
        const bool gain = r >= (uint32_t)kNumRes;
        const int i0 = gain ? 0 : 2 * (int)r, i1 = gain ? 1 : 2 * (int)r + 1;
        const double f = fade_value(fo[i0], fn[i0], ratio), bw = fade_value(fo[i1], fn[i1], ratio);
        const Coef k = resonator_coefficients_inline<MODE_EXACT>(f, bw, r == 0u, T.negPiOverSr, T.twoPiOverSr);
        double2 v = second ? make_double2(k.a, 0.0) : make_double2(k.b, k.c);
        if (gain) {
            int ga = 28, gb = -1;
#pragma unroll
            for (int g = kNumRes; g < kTrackEntries; ++g)
                if (r == (uint32_t)g) { ga = entry_value(g, 0); gb = entry_value(g, 1); }
            v.x = fade_value(fo[ga], fn[ga], ratio);
            v.y = gb >= 0 ? fade_value(fo[gb], fn[gb], ratio) : 0.0;
        }
        if (e0 + (uint32_t)lane < total) out[partAt + e] = v;
    }
    partAt += total;
    }
}

}  // namespace klatt
