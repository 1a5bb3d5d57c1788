/* speechPlayer_batch_digest's per-utterance value (csrc/klatt_engine.hip, pcm_digest: the sum over an utterance's samples of
 * mix64(position, value)) for PCM that lies in host memory -- the oracle's -- so that EVERY utterance of a full-size batch can be
 * compared with the oracle without copying the engine's PCM off the device.  Test infrastructure (tests/whole_batch.py builds it
 * with gcc -fopenmp); tests/test_gpu_parity.py checks this file's formula against numpy's restatement and the device's kernel. */
#include <stdint.h>

static inline uint64_t digest_mix(uint64_t pos, uint32_t value16)
{
    uint64_t x = (pos + 1ull) * 0x9E3779B97F4A7C15ull ^ ((uint64_t)value16 + 1ull) * 0xC2B2AE3D27D4EB4Full;
    x ^= x >> 29; x *= 0xBF58476D1CE4E5B9ull; x ^= x >> 32;
    return x;
}

/* utterance u: samples start[u] .. start[u + 1] - 1 of pcm */
void pcm_digest_many(const int16_t* pcm, const long long* start, long long nUtt, uint64_t* out)
{
#pragma omp parallel for schedule(dynamic, 16)
    for (long long u = 0; u < nUtt; ++u) {
        const int16_t* p = pcm + start[u];
        const long long n = start[u + 1] - start[u];
        uint64_t acc = 0;
        for (long long i = 0; i < n; ++i) acc += digest_mix((uint64_t)i, (uint16_t)p[i]);
        out[u] = acc;
    }
}
