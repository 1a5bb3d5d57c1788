// With two 4-wave workgroups resident per CU (80 KB of LDS each): which SIMD and which wave slot does wave w of
// each workgroup get?  Decides whether rotating the stage->wave map of every other workgroup can even out the
// per-SIMD load of the stage-parallel kernel.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <map>
#include <vector>
__global__ void probe(unsigned* out, unsigned long long hold)
{
    extern __shared__ unsigned char lds[];
    unsigned id = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));   // HW_REG_HW_ID
    unsigned xcc = __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (31 << 11)); // HW_REG_XCC_ID
    if ((threadIdx.x & 63) == 0) { out[(blockIdx.x * 4 + (threadIdx.x >> 6)) * 2] = id; out[(blockIdx.x * 4 + (threadIdx.x >> 6)) * 2 + 1] = xcc; }
    lds[threadIdx.x] = (unsigned char)id;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    while (__builtin_amdgcn_s_memtime() - t0 < hold) {}
}
int main()
{
    const int grid = 1536;   // 512 resident at once (2 per CU), then a second and third round as slots free up
    unsigned* d; hipMalloc(&d, grid * 4 * 2 * 4);
    hipFuncSetAttribute((const void*)probe, hipFuncAttributeMaxDynamicSharedMemorySize, 81920);
    hipLaunchKernelGGL(probe, dim3(grid), dim3(256), 81680, 0, d, 400000ull);
    hipDeviceSynchronize();
    std::vector<unsigned> h(grid * 8);
    hipMemcpy(h.data(), d, h.size() * 4, hipMemcpyDeviceToHost);
    int identity = 0, rotated = 0, other = 0, slotsSame = 0;
    std::map<unsigned, std::vector<int>> perCu;
    for (int b = 0; b < grid; ++b) {
        unsigned simd[4], slot[4];
        for (int w = 0; w < 4; ++w) { unsigned id = h[(b * 4 + w) * 2]; simd[w] = (id >> 4) & 3; slot[w] = id & 15; }
        const bool rot = (simd[1] == ((simd[0] + 1) & 3)) && (simd[2] == ((simd[0] + 2) & 3)) && (simd[3] == ((simd[0] + 3) & 3));
        if (rot && simd[0] == 0) identity++; else if (rot) rotated++; else other++;
        if (slot[0] == slot[1] && slot[1] == slot[2] && slot[2] == slot[3]) slotsSame++;
        unsigned id = h[b * 8], xcc = h[b * 8 + 1];
        perCu[(xcc & 15) << 16 | ((id >> 8) & 0xFF) << 4 | ((id >> 13) & 7) << 12] .push_back(b);
    }
    printf("workgroups: %d; wave w on SIMD w: %d; on SIMD (w + k) mod 4, k != 0: %d; other: %d; same wave slot on all four SIMDs: %d\n", grid, identity, rotated, other, slotsSame);
    printf("distinct CUs seen: %zu\n", perCu.size());
    int shown = 0;
    for (auto& kv : perCu) {
        if (shown++ >= 6) break;
        printf("cu key %06x:", kv.first);
        for (int b : kv.second) {
            printf("  wg%-4d simd[%u%u%u%u] slot[%u%u%u%u]", b, (h[b * 8] >> 4) & 3, (h[b * 8 + 2] >> 4) & 3, (h[b * 8 + 4] >> 4) & 3, (h[b * 8 + 6] >> 4) & 3,
                   h[b * 8] & 15, h[b * 8 + 2] & 15, h[b * 8 + 4] & 15, h[b * 8 + 6] & 15);
        }
        printf("\n");
    }
    return 0;
}
