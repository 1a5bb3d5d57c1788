// Where do the 4 wavefronts of a 256-thread workgroup land?  Prints SIMD / CU ids per wave (HW_REG_HW_ID).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void probe(unsigned* out)
{
    unsigned id = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));   // HW_REG_HW_ID, offset 0, size 32
    unsigned xcc = __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (31 << 11)); // HW_REG_XCC_ID
    if ((threadIdx.x & 63) == 0) { out[(blockIdx.x * 4 + (threadIdx.x >> 6)) * 2] = id; out[(blockIdx.x * 4 + (threadIdx.x >> 6)) * 2 + 1] = xcc; }
    // keep the workgroup resident for a while so that blocks spread over CUs
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    while (__builtin_amdgcn_s_memtime() - t0 < 200000) {}
}
int main()
{
    const int grid = 64;
    unsigned* d; hipMalloc(&d, grid * 4 * 2 * 4);
    probe<<<grid, 256>>>(d);
    hipDeviceSynchronize();
    std::vector<unsigned> h(grid * 8);
    hipMemcpy(h.data(), d, h.size() * 4, hipMemcpyDeviceToHost);
    for (int b = 0; b < 12; ++b) {
        printf("block %2d:", b);
        for (int w = 0; w < 4; ++w) {
            unsigned id = h[(b * 4 + w) * 2], xcc = h[(b * 4 + w) * 2 + 1];
            printf("  [wave%d simd=%u cu=%u sh=%u se=%u xcc=%u]", id & 15, (id >> 4) & 3, (id >> 8) & 15, (id >> 12) & 1, (id >> 13) & 7, xcc & 15);
        }
        printf("\n");
    }
    return 0;
}
