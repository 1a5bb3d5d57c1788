// Probe: what does it cost to hand a stream of f64 samples from a wavefront on one CU to a wavefront on
// ANOTHER CU through a ring in global memory (agent-scope release/acquire flags)?  This is the link a
// pipeline that spans several workgroups per 64 utterances would need (a 4096-utterance batch fills only
// 64 of 256 CUs with one workgroup per 64 utterances).
//
//   producer wave: per sample `OPS` dependent f64 operations, store [block % R][sample][lane]; per block
//                  of B samples: wait for room, then release-store head
//   consumer wave: wait for head, acquire, per sample load + `OPS` dependent f64 operations; per block
//                  release-store tail
// Every spin is bounded: a wave that waits too long raises `abort` and every wave leaves.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

constexpr int kFlagStride = 32;   // uint32 per flag slot: 128 bytes apart

__device__ __forceinline__ bool wait_ge(unsigned* flag, unsigned want, unsigned* abortFlag)
{
    for (int spin = 0; spin < (1 << 22); ++spin) {
        const unsigned v = __hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if ((int)(v - want) >= 0) return true;                    // the caller issues the agent-scope acquire fence
        if ((spin & 63) == 63 && __hip_atomic_load(abortFlag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return false;
        __builtin_amdgcn_s_sleep(2);
    }
    __hip_atomic_store(abortFlag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return false;
}

template <int OPS, int B, int R, bool SYNC>
__global__ void __launch_bounds__(256) ring(double* rings, unsigned* flags, double* out, int nBlocks, int G, int adjacent, unsigned* abortFlag)
{
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int group = adjacent ? (blockIdx.x >> 1) : (blockIdx.x % G);
    const int role = adjacent ? (blockIdx.x & 1) : (blockIdx.x / G);
    double* const myRing = rings + (size_t)group * R * B * 64;
    unsigned* const head = flags + (size_t)group * 2 * kFlagStride;
    unsigned* const tail = head + kFlagStride;
    const double a = 0.01 + 1e-4 * lane, b = 0.999 - 1e-4 * lane;
    if (role == 0 && wave == 3) {
        double z = 1.0 + lane;
        for (int blk = 0; blk < nBlocks; ++blk) {
            if (SYNC && blk >= R) { if (!wait_ge(tail, (unsigned)(blk - R + 1), abortFlag)) return; }
            double* dst = myRing + (size_t)(blk % R) * B * 64 + lane;
#pragma unroll 8
            for (int i = 0; i < B; ++i) {
#pragma unroll
                for (int k = 0; k < OPS; ++k) z = z * b + a;      // unfused build: 2 dependent ops per k
                dst[i * 64] = z;
            }
            if (SYNC) {
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                if (lane == 0) __hip_atomic_store(head, (unsigned)(blk + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        out[blockIdx.x * 64 + lane] = z;
    } else if (role == 1 && wave == 0) {
        double z = 0.0;
        for (int blk = 0; blk < nBlocks; ++blk) {
            if (SYNC) {
                if (!wait_ge(head, (unsigned)(blk + 1), abortFlag)) return;
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            }
            const double* src = myRing + (size_t)(blk % R) * B * 64 + lane;
#pragma unroll 8
            for (int i = 0; i < B; ++i) {
                double v = __builtin_nontemporal_load(src + i * 64);
                z = z + v;
#pragma unroll
                for (int k = 0; k < OPS; ++k) z = z * b + a;
            }
            if (SYNC) {
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                if (lane == 0) __hip_atomic_store(tail, (unsigned)(blk + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        out[blockIdx.x * 64 + lane] = z;
    }
}

template <int OPS, int B, int R, bool SYNC>
int run(int G, int adjacent, double* rings, unsigned* flags, double* out, unsigned* abortFlag)
{
    const int samples = 22050;
    const int nBlocks = (samples + B - 1) / B;
    float best = 1e9f;
    for (int rep = 0; rep < 3; ++rep) {
        CHECK(hipMemset(flags, 0, 1 << 20));
        CHECK(hipMemset(abortFlag, 0, 4));
        hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
        CHECK(hipEventRecord(e0));
        ring<OPS, B, R, SYNC><<<2 * G, 256>>>(rings, flags, out, nBlocks, G, adjacent, abortFlag);
        CHECK(hipEventRecord(e1));
        CHECK(hipDeviceSynchronize());
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
        CHECK(hipEventDestroy(e0)); CHECK(hipEventDestroy(e1));
    }
    unsigned ab = 0; CHECK(hipMemcpy(&ab, abortFlag, 4, hipMemcpyDeviceToHost));
    printf("ops/sample=%2d block=%4d ring=%d sync=%d pairs=%3d %s: %7.3f ms = %6.1f ns per sample%s\n", 2 * OPS, B, R, (int)SYNC, G,
           adjacent ? "adjacent WGs" : "WGs g, g+G  ", best, best * 1e6 / (nBlocks * B), ab ? "  ABORTED (spin limit)" : "");
    return 0;
}

int main()
{
    double* rings; unsigned* flags; double* out; unsigned* abortFlag;
    CHECK(hipMalloc(&rings, (size_t)256 * 4 * 512 * 64 * 8));
    CHECK(hipMalloc(&flags, 1 << 20));
    CHECK(hipMalloc(&out, 1 << 20));
    CHECK(hipMalloc(&abortFlag, 4));
    for (int adjacent : {0, 1}) {
        for (int G : {64, 128}) {
            run<5, 128, 4, false>(G, adjacent, rings, flags, out, abortFlag);
            run<5, 128, 4, true>(G, adjacent, rings, flags, out, abortFlag);
            run<5, 64, 4, true>(G, adjacent, rings, flags, out, abortFlag);
            run<5, 32, 8, true>(G, adjacent, rings, flags, out, abortFlag);
            run<5, 256, 4, true>(G, adjacent, rings, flags, out, abortFlag);
            run<3, 128, 4, true>(G, adjacent, rings, flags, out, abortFlag);
            run<8, 128, 4, true>(G, adjacent, rings, flags, out, abortFlag);
        }
    }
    return 0;
}
