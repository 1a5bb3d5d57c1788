// Probe: cost of the resonator recurrence y = a*x + b*z1 + c*z2 (unfused and fused) with every operand in
// a VGPR, as a serial chain of R resonators -- the inner pattern of the Klatt kernels.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int R, bool FUSED, bool LDSIO>
__global__ void chain(double* out, int iters, double seed)
{
    __shared__ double pipe[2][16][64];
    const int lane = threadIdx.x & 63;
    double a[R], b[R], c[R], z1[R], z2[R];
#pragma unroll
    for (int r = 0; r < R; ++r) { a[r] = 0.01 + 1e-4 * (lane + r); b[r] = 1.9 - 1e-3 * (lane + r); c[r] = -0.95 + 1e-4 * r; z1[r] = 0; z2[r] = 0; }
    double x = seed + lane * 1e-3;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            double v = LDSIO ? pipe[it & 1][i][lane] + x : x;
#pragma unroll
            for (int r = 0; r < R; ++r) {
                double y = FUSED ? __builtin_fma(c[r], z2[r], __builtin_fma(b[r], z1[r], a[r] * v)) : a[r] * v + b[r] * z1[r] + c[r] * z2[r];
                z2[r] = z1[r]; z1[r] = y; v = y;
            }
            if (LDSIO) pipe[(it + 1) & 1][i][lane] = v; else x = v * 1e-3 + seed;
        }
        if (LDSIO) __syncthreads();
    }
    double s = x;
#pragma unroll
    for (int r = 0; r < R; ++r) s += z1[r] + z2[r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int R, bool FUSED, bool LDSIO>
int run(const char* name, int block, int grid, double* d)
{
    const int iters = 2000;
    chain<R, FUSED, LDSIO><<<grid, block>>>(d, 10, 1.0);
    CHECK(hipDeviceSynchronize());
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    CHECK(hipEventRecord(e0));
    chain<R, FUSED, LDSIO><<<grid, block>>>(d, iters, 1.0);
    CHECK(hipEventRecord(e1));
    CHECK(hipDeviceSynchronize());
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    double nsPerSample = ms * 1e6 / (iters * 16.0);
    printf("%-34s block=%3d grid=%3d: %7.2f ns per sample per wave (%5.2f ns per f64 op, %d ops/sample)\n", name, block, grid, nsPerSample,
           nsPerSample / (R * (FUSED ? 3 : 5)), R * (FUSED ? 3 : 5));
    return 0;
}

// Same arithmetic, software-pipelined by hand: at step i resonator r works on sample i - r, so the R
// updates of one step are independent of each other (they only read the previous step's links).
template <int R, int CH>
__global__ void chain_skewed(double* out, int iters, double seed)
{
    __shared__ double pipe[2][CH][64];
    const int lane = threadIdx.x & 63;
    double a[R], b[R], c[R], z1[R], z2[R];
#pragma unroll
    for (int r = 0; r < R; ++r) { a[r] = 0.01 + 1e-4 * (lane + r); b[r] = 1.9 - 1e-3 * (lane + r); c[r] = -0.95 + 1e-4 * r; z1[r] = 0; z2[r] = 0; }
    double x = seed + lane * 1e-3;
    for (int it = 0; it < iters; ++it) {
        double in[CH];
#pragma unroll
        for (int i = 0; i < CH; ++i) in[i] = pipe[it & 1][i][lane] + x;
        double link[R + 1];
#pragma unroll
        for (int i = 0; i < CH + R - 1; ++i) {
            double nl[R + 1];
            if (i < CH) nl[0] = in[i];
#pragma unroll
            for (int r = 0; r < R; ++r) {
                if (i - r >= 0 && i - r < CH) {
                    const double v = (r == 0) ? in[i] : link[r];
                    const double y = a[r] * v + b[r] * z1[r] + c[r] * z2[r];
                    z2[r] = z1[r]; z1[r] = y;
                    nl[r + 1] = y;
                    if (r == R - 1) pipe[(it + 1) & 1][i - r][lane] = y;
                }
            }
#pragma unroll
            for (int r = 1; r <= R; ++r) link[r] = nl[r];
        }
        __syncthreads();
    }
    double s = x;
#pragma unroll
    for (int r = 0; r < R; ++r) s += z1[r] + z2[r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int R, int CH>
int run_skewed(const char* name, int block, int grid, double* d)
{
    const int iters = 2000 * 16 / CH;
    chain_skewed<R, CH><<<grid, block>>>(d, 10, 1.0);
    CHECK(hipDeviceSynchronize());
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    CHECK(hipEventRecord(e0));
    chain_skewed<R, CH><<<grid, block>>>(d, iters, 1.0);
    CHECK(hipEventRecord(e1));
    CHECK(hipDeviceSynchronize());
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    double nsPerSample = ms * 1e6 / (iters * (double)CH);
    printf("%-34s block=%3d grid=%3d: %7.2f ns per sample per wave (%5.2f ns per f64 op, %d ops/sample)\n", name, block, grid, nsPerSample,
           nsPerSample / (R * 5), R * 5);
    return 0;
}

int main()
{
    double* d; CHECK(hipMalloc(&d, 1 << 24));
    for (int grid : {64, 256}) {
        run<3, false, false>("3 resonators unfused, regs", 64, grid, d);
        run<3, true, false>("3 resonators fused, regs", 64, grid, d);
        run<8, false, false>("8 resonators unfused, regs", 64, grid, d);
        run<8, true, false>("8 resonators fused, regs", 64, grid, d);
        run<3, false, true>("3 resonators unfused, LDS io+barrier", 64, grid, d);
        run<3, false, true>("3 resonators unfused, LDS io+barrier", 256, grid, d);
        run<3, true, true>("3 resonators fused, LDS io+barrier", 256, grid, d);
        run<3, false, false>("3 resonators unfused, regs", 256, grid, d);
        run_skewed<3, 16>("3 res unfused SKEWED ch16, LDS io", 64, grid, d);
        run_skewed<3, 16>("3 res unfused SKEWED ch16, LDS io", 256, grid, d);
        run_skewed<3, 32>("3 res unfused SKEWED ch32, LDS io", 256, grid, d);
        run_skewed<5, 16>("5 res unfused SKEWED ch16, LDS io", 256, grid, d);
    }
    return 0;
}
