// Issue-rate probes that size the Klatt kernel's design (results quoted in DESIGN.md).
// Each kernel runs N iterations of 8 independent dependency chains per lane and reports
// shader cycles (s_memtime) per wave-instruction, for 1, 2 or 4 waves per SIMD and for
// 64 or 16 active lanes.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int KIND>
__global__ void probe(unsigned long long* out, int iters, int activeLanes, double seedd)
{
    const int lane = threadIdx.x & 63;
    double a0 = seedd + lane, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    float f0 = (float)a0, f1 = (float)a1, f2 = (float)a2, f3 = (float)a3, f4 = (float)a4, f5 = (float)a5, f6 = (float)a6, f7 = (float)a7;
    unsigned u0 = lane + 1, u1 = lane + 2, u2 = lane + 3, u3 = lane + 4, u4 = lane + 5, u5 = lane + 6, u6 = lane + 7, u7 = lane + 8;
    const double m = 0.999999, c = 1e-9;
    const float mf = 0.999999f, cf = 1e-9f;
    unsigned long long t0 = 0, t1 = 0;
    if (lane < activeLanes) {
        t0 = __builtin_amdgcn_s_memtime();
        for (int i = 0; i < iters; ++i) {
#pragma unroll
          for (int rep = 0; rep < 8; ++rep) {
            if (KIND == 0) {  // v_fma_f64
                a0 = __builtin_fma(a0, m, c); a1 = __builtin_fma(a1, m, c); a2 = __builtin_fma(a2, m, c); a3 = __builtin_fma(a3, m, c);
                a4 = __builtin_fma(a4, m, c); a5 = __builtin_fma(a5, m, c); a6 = __builtin_fma(a6, m, c); a7 = __builtin_fma(a7, m, c);
            } else if (KIND == 1) {  // v_fma_f32
                f0 = __builtin_fmaf(f0, mf, cf); f1 = __builtin_fmaf(f1, mf, cf); f2 = __builtin_fmaf(f2, mf, cf); f3 = __builtin_fmaf(f3, mf, cf);
                f4 = __builtin_fmaf(f4, mf, cf); f5 = __builtin_fmaf(f5, mf, cf); f6 = __builtin_fmaf(f6, mf, cf); f7 = __builtin_fmaf(f7, mf, cf);
            } else if (KIND == 2) {  // v_mul_lo_u32
                u0 *= 0x7FEB352Du; u1 *= 0x7FEB352Du; u2 *= 0x7FEB352Du; u3 *= 0x7FEB352Du;
                u4 *= 0x7FEB352Du; u5 *= 0x7FEB352Du; u6 *= 0x7FEB352Du; u7 *= 0x7FEB352Du;
            } else if (KIND == 11) { // v_mul_lo_u32 by a value the compiler cannot fold (KIND 2 folds its eight constant multiplies into one)
                u0 *= u1 | 1u; u1 *= u2 | 1u; u2 *= u3 | 1u; u3 *= u0 | 1u; u4 *= u5 | 1u; u5 *= u6 | 1u; u6 *= u7 | 1u; u7 *= u4 | 1u;
            } else if (KIND == 12) { // the v_or_b32 of KIND 11 alone with an add in the multiply's place: what to subtract
                u0 += u1 | 1u; u1 += u2 | 1u; u2 += u3 | 1u; u3 += u0 | 1u; u4 += u5 | 1u; u5 += u6 | 1u; u6 += u7 | 1u; u7 += u4 | 1u;
            } else if (KIND == 3) {  // v_mul_f64 + v_add_f64 (unfused)
                a0 = a0 * m; a1 = a1 * m; a2 = a2 * m; a3 = a3 * m; a0 = a0 + c; a1 = a1 + c; a2 = a2 + c; a3 = a3 + c;
            } else if (KIND == 4) {  // dependent v_fma_f64 chain (latency)
                a0 = __builtin_fma(a0, m, c); a0 = __builtin_fma(a0, m, c); a0 = __builtin_fma(a0, m, c); a0 = __builtin_fma(a0, m, c);
                a0 = __builtin_fma(a0, m, c); a0 = __builtin_fma(a0, m, c); a0 = __builtin_fma(a0, m, c); a0 = __builtin_fma(a0, m, c);
            } else if (KIND == 5) {  // dependent v_fma_f32 chain (latency)
                f0 = __builtin_fmaf(f0, mf, cf); f0 = __builtin_fmaf(f0, mf, cf); f0 = __builtin_fmaf(f0, mf, cf); f0 = __builtin_fmaf(f0, mf, cf);
                f0 = __builtin_fmaf(f0, mf, cf); f0 = __builtin_fmaf(f0, mf, cf); f0 = __builtin_fmaf(f0, mf, cf); f0 = __builtin_fmaf(f0, mf, cf);
            } else if (KIND == 7) {   // ONE dependent chain of alternating v_mul_f64 / v_add_f64 (8 instructions)
                a0 = a0 * m; a0 = a0 + c; a0 = a0 * m; a0 = a0 + c; a0 = a0 * m; a0 = a0 + c; a0 = a0 * m; a0 = a0 + c;
            } else if (KIND == 8) {   // TWO such chains interleaved
                a0 = a0 * m; a1 = a1 * m; a0 = a0 + c; a1 = a1 + c; a0 = a0 * m; a1 = a1 * m; a0 = a0 + c; a1 = a1 + c;
            } else if (KIND == 9) {   // FOUR such chains interleaved
                a0 = a0 * m; a1 = a1 * m; a2 = a2 * m; a3 = a3 * m; a0 = a0 + c; a1 = a1 + c; a2 = a2 + c; a3 = a3 + c;
            } else if (KIND == 10) {  // dependent v_add_f64 chain
                a0 = a0 + c; a0 = a0 + m; a0 = a0 + c; a0 = a0 + m; a0 = a0 + c; a0 = a0 + m; a0 = a0 + c; a0 = a0 + m;
            } else if (KIND == 6) {  // v_pk_fma_f32 (two floats per lane per instruction)
                typedef float v2 __attribute__((ext_vector_type(2)));
                v2 p0 = {f0, f1}, p1 = {f2, f3}, p2 = {f4, f5}, p3 = {f6, f7};
                v2 mm = {mf, mf}, cc = {cf, cf};
                for (int k = 0; k < 2; ++k) {
                    p0 = __builtin_elementwise_fma(p0, mm, cc); p1 = __builtin_elementwise_fma(p1, mm, cc);
                    p2 = __builtin_elementwise_fma(p2, mm, cc); p3 = __builtin_elementwise_fma(p3, mm, cc);
                }
                f0 = p0.x; f1 = p0.y; f2 = p1.x; f3 = p1.y; f4 = p2.x; f5 = p2.y; f6 = p3.x; f7 = p3.y;
            }
          }
        }
        t1 = __builtin_amdgcn_s_memtime();
    }
    double sink = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + f0 + f1 + f2 + f3 + f4 + f5 + f6 + f7 + (double)(u0 ^ u1 ^ u2 ^ u3 ^ u4 ^ u5 ^ u6 ^ u7);
    if (sink == 12345.678) out[1000000] = 1;  // never true; keeps the chains alive
    if (lane == 0) out[blockIdx.x * (blockDim.x / 64) + (threadIdx.x >> 6)] = t1 - t0;
}

template <int KIND>
int run(const char* name, int wavesPerSimd, int activeLanes, unsigned long long* dOut)
{
    const int iters = 4000;
    const int block = 64 * 4 * wavesPerSimd;  // one workgroup fills every SIMD of its CU with wavesPerSimd waves
    const int grid = 256;
    probe<KIND><<<grid, block>>>(dOut, 10, activeLanes, 1.0);
    CHECK(hipDeviceSynchronize());
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    CHECK(hipEventRecord(e0));
    probe<KIND><<<grid, block>>>(dOut, iters, activeLanes, 1.0);
    CHECK(hipEventRecord(e1));
    CHECK(hipDeviceSynchronize());
    float msec = 0; CHECK(hipEventElapsedTime(&msec, e0, e1));
    std::vector<unsigned long long> h(grid * block / 64);
    CHECK(hipMemcpy(h.data(), dOut, h.size() * 8, hipMemcpyDeviceToHost));
    double mean = 0;
    for (auto v : h) mean += (double)v;
    mean /= h.size();
    // s_memtime ticks at 100 MHz on this part? report both raw ticks and per-instruction
    const double instr = (double)iters * 64.0;   // wave-instructions per wave
    const double nsPerInstrPerSimd = msec * 1e6 / (instr * wavesPerSimd);
    printf("%-22s waves/SIMD=%d lanes=%2d : %6.3f memtime ticks/instr/wave, %6.3f ns per instr per SIMD (=%5.2f cyc @2.4GHz), kernel %.3f ms\n",
           name, wavesPerSimd, activeLanes, mean / instr, nsPerInstrPerSimd, nsPerInstrPerSimd * 2.4, msec);
    return 0;
}

int main()
{
    unsigned long long* dOut;
    CHECK(hipMalloc(&dOut, 1000008 * 8));
    hipDeviceProp_t p;
    CHECK(hipGetDeviceProperties(&p, 0));
    printf("%s, %d CUs, clock %d kHz\n", p.name, p.multiProcessorCount, p.clockRate);
    for (int w : {1, 2}) {
        run<7>("mul/add f64, 1 dep chain", w, 64, dOut);
        run<8>("mul/add f64, 2 chains", w, 64, dOut);
        run<9>("mul/add f64, 4 chains", w, 64, dOut);
        run<10>("v_add_f64 dep chain", w, 64, dOut);
        run<4>("v_fma_f64 dep chain", w, 64, dOut);
    }
    for (int w : {1, 2, 4}) {
        run<0>("v_fma_f64", w, 64, dOut);
        run<0>("v_fma_f64", w, 16, dOut);
        run<3>("v_mul_f64+v_add_f64", w, 64, dOut);
        run<1>("v_fma_f32", w, 64, dOut);
        run<1>("v_fma_f32", w, 16, dOut);
        run<6>("v_pk_fma_f32", w, 64, dOut);
        run<2>("v_mul_lo_u32", w, 64, dOut);
        run<11>("v_or + v_mul_lo_u32", w, 64, dOut);
        run<12>("v_or + v_add_u32", w, 64, dOut);
        run<4>("v_fma_f64 dependent", w, 64, dOut);
        run<5>("v_fma_f32 dependent", w, 64, dOut);
    }
    return 0;
}
