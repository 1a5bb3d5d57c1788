// ubench_lanepipe.hip -- what does one step of the lane-pipelined resonator cost a lone wave?
// Variants of the hand-over: none, wave_ror:1, row_shr:1, ds_bpermute, with/without the select of the
// first lane's LDS input, with/without the LDS store.  hipcc -O3 --offload-arch=gfx950 -ffp-contract=off
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int CTRL>
__device__ __forceinline__ double dpp_move(double v)
{
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_mov_dpp(lo, CTRL, 0xF, 0xF, false);
    hi = __builtin_amdgcn_mov_dpp(hi, CTRL, 0xF, 0xF, false);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double bperm_move(double v, int src)
{
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_ds_bpermute(src * 4, lo);
    hi = __builtin_amdgcn_ds_bpermute(src * 4, hi);
    return __hiloint2double(hi, lo);
}

// VARIANT: 0 no hand-over (in = out); 1 wave_ror:1; 2 row_shr:1; 3 ds_bpermute; 4 wave_ror + select; 5 wave_ror + select + LDS store
//          6: two independent pipelines interleaved (variant 5 twice); 7: FMA form y = fma(a, in, s), s = fma(b, z1, c * z2) + select + store
// WORKMASK: which waves of the block work (the others leave at once); every working wave has its own LDS areas
template <int VARIANT, int NWAVES = 1, int WORKMASK = 1>
__global__ void __launch_bounds__(64 * NWAVES) k(double* out, const double* coef, int steps, int probe)
{
    __shared__ double xsAll[NWAVES][64 * 32];
    __shared__ double ysAll[NWAVES][64 * 33];
    if (!((WORKMASK >> (threadIdx.x >> 6)) & 1)) return;
    double* const xs = xsAll[threadIdx.x >> 6];
    double* const ys = ysAll[threadIdx.x >> 6];
    const int lane = threadIdx.x & 63;
    const double a = coef[lane], b = coef[64 + lane], c = coef[128 + lane];
    double z1 = 0.0, z2 = 0.0, y = 1e-3 * lane;
    double z1b = 0.0, z2b = 0.0, yb = 2e-3 * lane;
    const bool first = (lane % 6) == 0;
    for (int i = 0; i < 32; ++i) xs[i * 64 + lane] = 1e-3 * (i + lane);
    __builtin_amdgcn_s_waitcnt(0xC07F);      // this wave's own LDS writes (no barrier: some waves have left)
    const int src = (lane + 63) & 63;
    for (int t = 0; t < steps; t += 32) {
        double pre[32];
        // VARIANT 9 / 11: the real kernel's input rows (20 doubles apart, the 6 lanes of an utterance read one address);
        // VARIANT 10 / 11: its output rows (20 doubles apart; last lanes into the ring, the others into a scratch row)
        const bool lastL = (lane % 6) == 5;
        double* const yrow = lastL ? (ys + lane / 6) : (ys + 32 * 20 + lane);
#pragma unroll
        for (int i = 0; i < 32; ++i) pre[i] = (VARIANT == 9 || VARIANT == 11) ? xs[i * 20 + lane / 6] : xs[i * 64 + lane];
#pragma unroll
        for (int i = 0; i < 32; ++i) {
            double in;
            if (VARIANT == 0) in = y;
            else if (VARIANT == 2 || VARIANT == 8) in = dpp_move<0x111>(y);
            else if (VARIANT == 3) in = bperm_move(y, src);
            else in = dpp_move<0x13C>(y);
            if (VARIANT >= 4) in = first ? pre[i] : in;
            if (VARIANT == 7) {
                const double s = __builtin_fma(b, z1, c * z2);
                const double w = __builtin_fma(a, in, s);
                z2 = z1; z1 = w; y = w;
            } else {
                const double w = a * in + b * z1 + c * z2;
                z2 = z1; z1 = w; y = w;
            }
            if (VARIANT == 10 || VARIANT == 11) yrow[i * 20] = y;
            else if (VARIANT >= 5) ys[i * 66 + lane] = y;
            if (VARIANT == 6) {
                double inb = dpp_move<0x13C>(yb);
                inb = first ? pre[31 - i] : inb;
                const double w = a * inb + b * z1b + c * z2b;
                z2b = z1b; z1b = w; yb = w;
                ys[i * 66 + 2 + lane] = yb;
            }
        }
    }
    out[blockIdx.x * 64 * NWAVES + threadIdx.x] = y + yb + ys[(probe & 31) * 66 + lane];
}

// The whole workgroup of klatt_lanepipe.h in miniature: wave 0 source-like (phase chain + 5 elementwise operations
// + LDS store), waves 1-2 the filter step of variant 5 fed from LDS, wave 3 final-like (two multiplies, clamp,
// convert, ds_write_b16), one barrier per 32-sample chunk, double-buffered pipes.  ROLES: bit w set = wave w works.
// CUT (timing only): 1 the filter waves load nothing, 2 store nothing, 4 no select of the first lane
template <int ROLES, bool SYNC = true, int CUT = 0>
__global__ void __launch_bounds__(256) wg(double* out, const double* coef, int steps)
{
    __shared__ double px[2 * 32 * 20];
    __shared__ double py[4 * 32 * 20 + 64 + 32 * 20];
    __shared__ short tile[64 * 36];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const double a = coef[lane], b = coef[64 + lane], c = coef[128 + lane];
    double z1 = 0.0, z2 = 0.0, y = 1e-3 * lane, pp = 0.0, acc = 0.0;
    const bool first = (lane % 6) == 0, last = (lane % 6) == 5;
    const int uw = (wave == 1 || wave == 2) ? ((wave - 1) * 10 + lane / 6) % 20 : lane % 20;
    for (int i = threadIdx.x; i < 2 * 32 * 20; i += 256) px[i] = 1e-3 * i;
    for (int i = threadIdx.x; i < 4 * 32 * 20; i += 256) py[i] = 1e-3 * i;
    __syncthreads();
    const int nChunks = steps / 32;
    for (int it = 0; it < nChunks + 3; ++it) {
        if (wave == 0 && (ROLES & 1)) {
            double* row = px + ((it & 1) * 32) * 20 + uw;
            if (lane < 20) {
#pragma unroll
                for (int i = 0; i < 32; ++i) {
                    const double t = b + pp;
                    pp = t - __builtin_trunc(t);
                    row[i * 20] = ((((pp * 2.0) - 1.0) * a) * c) * 0.5;
                }
            }
        } else if ((wave == 1 || wave == 2) && (ROLES & (1 << wave))) {
            const double* xin = px + (((it + 1) & 1) * 32) * 20 + uw;
            double* yrow = last ? (py + (((it + 3) & 3) * 32) * 20 + uw) : (py + 4 * 32 * 20 + lane);
            double pre[32];
#pragma unroll
            for (int i = 0; i < 32; ++i) pre[i] = (CUT & 1) ? 0.001 * i : xin[i * 20];
#pragma unroll
            for (int i = 0; i < 32; ++i) {
                double in = dpp_move<0x13C>(y);
                if (!(CUT & 4)) in = first ? pre[i] : in; else in += pre[i];
                const double w = a * in + b * z1 + c * z2;
                z2 = z1; z1 = w; y = w;
                if (!(CUT & 2)) yrow[i * 20] = y;
            }
        } else if (wave == 3 && (ROLES & 8)) {
            const double* yin = py + uw;
            short* myRow = tile + lane * 36;
            if (lane < 20) {
                double pre[32];
#pragma unroll
                for (int i = 0; i < 32; ++i) pre[i] = yin[((((it + 1) & 3) * 32 + i + 5) & 127) * 20];
#pragma unroll
                for (int i = 0; i < 32; ++i) {
                    const double v = (pre[i] * a) * 4000.0;
                    const double lo = (v < 32000.0) ? v : 32000.0;
                    const double cl = (lo > -32000.0) ? lo : -32000.0;
                    myRow[i] = (short)(int)cl;
                }
                acc += myRow[5];
            }
        }
        if (SYNC) __syncthreads();
    }
    out[blockIdx.x * 256 + threadIdx.x] = y + pp + acc;
}

template <int ROLES, bool SYNC = true, int CUT = 0>
void run_wg(const char* name, double* dOut, double* dCoef, int blocks)
{
    const int steps = 32 * 1024;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((wg<ROLES, SYNC, CUT>), dim3(blocks), dim3(256), 0, 0, dOut, dCoef, steps);
    hipEventRecord(e0);
    hipLaunchKernelGGL((wg<ROLES, SYNC, CUT>), dim3(blocks), dim3(256), 0, 0, dOut, dCoef, steps);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    printf("%-58s blocks=%5d : %7.2f ns per step\n", name, blocks, ms * 1e6 / steps);
}

// What does an LDS store cost the issuing wave?  A resonator chain (5 f64 operations per step) plus, per step:
// KIND 0 nothing, 1 ds_write_b64, 2 ds_write_b32, 3 ds_write_b16, 4 one ds_write_b64 every 4th step,
// 5 ds_write_b64 from 16 lanes only, 6 one ds_write_b128 every 2nd step, 7 ds_read_b64 (result used 8 steps later)
template <int KIND>
__global__ void __launch_bounds__(64) ldscost(double* out, const double* coef, int steps, int probe)
{
    __shared__ double buf[64 * 40];
    const int lane = threadIdx.x;
    const double a = coef[lane], b = coef[64 + lane], c = coef[128 + lane];
    double z1 = 0.0, z2 = 0.0, y = 1e-3 * lane, acc = 0.0;
    for (int i = 0; i < 40; ++i) buf[i * 64 + lane] = 1e-3 * i;
    __syncthreads();
    double hold = 0.0;
    for (int t = 0; t < steps; t += 32) {
        double rd[32];
#pragma unroll
        for (int i = 0; i < 32; ++i) {
            double in = y + acc;
            const double w = a * in + b * z1 + c * z2;
            z2 = z1; z1 = w; y = w;
            if (KIND == 1) buf[i * 64 + lane] = y;
            if (KIND == 2) reinterpret_cast<float*>(buf)[i * 64 + lane] = (float)y;
            if (KIND == 3) reinterpret_cast<short*>(buf)[i * 66 + lane * 36] = (short)(int)y;
            if (KIND == 4 && (i & 3) == 3) buf[i * 64 + lane] = y;
            if (KIND == 5 && lane < 16) buf[i * 64 + lane] = y;
            if (KIND == 6) { if (i & 1) *reinterpret_cast<double2*>(&buf[(i >> 1) * 128 + lane * 2]) = make_double2(hold, y); else hold = y; }
            if (KIND == 7) { rd[i] = buf[i * 64 + lane]; if (i >= 8) acc = rd[i - 8] * 1e-30; }
        }
    }
    out[blockIdx.x * 64 + lane] = y + buf[(probe & 31) * 64 + lane];
}

template <int KIND>
void run_lds(const char* name, double* dOut, double* dCoef, int blocks)
{
    const int steps = 32 * 1024;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(ldscost<KIND>, dim3(blocks), dim3(64), 0, 0, dOut, dCoef, steps, 3);
    hipEventRecord(e0);
    hipLaunchKernelGGL(ldscost<KIND>, dim3(blocks), dim3(64), 0, 0, dOut, dCoef, steps, 3);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    printf("%-58s blocks=%5d : %7.2f ns per step\n", name, blocks, ms * 1e6 / steps);
}

// One stage of the stage-parallel kernel: three resonators in series per sample, input preloaded from LDS, output to LDS.
// SKEW 0: sample by sample (r_a, r_b, r_c of sample i back to back, as klatt_systolic.h's steady loop is written);
// SKEW 1: software-pipelined -- iteration i runs r_a on sample i, r_b on sample i - 1, r_c on sample i - 2, so the three
//         chains of an iteration are independent (same operations on the same operands, only issued in another order)
template <int SKEW>
__global__ void __launch_bounds__(64) chain3(double* out, const double* coef, int steps, int probe)
{
    __shared__ double xs[64 * 32];
    __shared__ double ys[64 * 32];
    const int lane = threadIdx.x;
    const double a0 = coef[lane], b0 = coef[64 + lane], c0 = coef[128 + lane];
    const double a1 = a0 * 1.01, b1 = b0 * 0.99, c1 = c0 * 1.02, a2 = a0 * 0.98, b2 = b0 * 1.01, c2 = c0 * 0.97;
    double p0 = 0, q0 = 0, p1 = 0, q1 = 0, p2 = 0, q2 = 0, oa = 0, ob = 0;
    for (int i = 0; i < 32; ++i) xs[i * 64 + lane] = 1e-3 * (i + lane);
    __syncthreads();
    for (int t = 0; t < steps; t += 32) {
        double pre[32];
#pragma unroll
        for (int i = 0; i < 32; ++i) pre[i] = xs[i * 64 + lane];
#pragma unroll
        for (int i = 0; i < 32; ++i) {
            if (SKEW == 0) {
                double o = a0 * pre[i] + b0 * p0 + c0 * q0; q0 = p0; p0 = o;
                o = a1 * o + b1 * p1 + c1 * q1; q1 = p1; p1 = o;
                o = a2 * o + b2 * p2 + c2 * q2; q2 = p2; p2 = o;
                ys[i * 64 + lane] = o;
            } else {
                // interleaved by hand: the products first, then the sums level by level
                const double ta = a0 * pre[i], tb = a1 * oa, tc = a2 * ob;
                const double ua = b0 * p0, ub = b1 * p1, uc = b2 * p2;
                const double va = c0 * q0, vb = c1 * q1, vc = c2 * q2;
                const double sa = ta + ua, sb = tb + ub, sc = tc + uc;
                const double na = sa + va, nb = sb + vb, nc = sc + vc;
                q0 = p0; p0 = na; q1 = p1; p1 = nb; q2 = p2; p2 = nc;
                oa = na; ob = nb;
                ys[i * 64 + lane] = nc;
            }
        }
    }
    out[blockIdx.x * 64 + lane] = p2 + ys[(probe & 31) * 64 + lane];
}

template <int SKEW>
void run_chain3(const char* name, double* dOut, double* dCoef, int blocks)
{
    const int steps = 32 * 1024;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(chain3<SKEW>, dim3(blocks), dim3(64), 0, 0, dOut, dCoef, steps, 3);
    hipEventRecord(e0);
    hipLaunchKernelGGL(chain3<SKEW>, dim3(blocks), dim3(64), 0, 0, dOut, dCoef, steps, 3);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    printf("%-58s blocks=%5d : %7.2f ns per step\n", name, blocks, ms * 1e6 / steps);
}

template <int V, int NWAVES, int WORKMASK>
void run_multi(const char* name, double* dOut, double* dCoef, int blocks)
{
    const int steps = 32 * 1024;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<V, NWAVES, WORKMASK>), dim3(blocks), dim3(64 * NWAVES), 0, 0, dOut, dCoef, steps, 3);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<V, NWAVES, WORKMASK>), dim3(blocks), dim3(64 * NWAVES), 0, 0, dOut, dCoef, steps, 3);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    printf("%-58s blocks=%5d : %7.2f ns per step\n", name, blocks, ms * 1e6 / steps);
}

template <int V>
void run(const char* name, double* dOut, double* dCoef, int blocks)
{
    const int steps = 32 * 1024;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<V>, dim3(blocks), dim3(64), 0, 0, dOut, dCoef, steps, 3);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<V>, dim3(blocks), dim3(64), 0, 0, dOut, dCoef, steps, 3);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    printf("%-58s blocks=%5d : %7.2f ns per step\n", name, blocks, ms * 1e6 / steps);
}

int main()
{
    double *dOut, *dCoef;
    hipMalloc(&dOut, 4096 * 256 * 8);
    hipMalloc(&dCoef, 192 * 8);
    std::vector<double> h(192);
    for (int i = 0; i < 64; ++i) { h[i] = 0.01; h[64 + i] = 1.2; h[128 + i] = -0.5; }
    hipMemcpy(dCoef, h.data(), 192 * 8, hipMemcpyHostToDevice);
    for (int blocks : {256, 1024, 2048}) {
        run_chain3<0>("three resonators in series, sample by sample", dOut, dCoef, blocks);
        run_chain3<1>("three resonators, software-pipelined (skewed)", dOut, dCoef, blocks);
    }
    for (int blocks : {256}) {
        run_lds<0>("chain only", dOut, dCoef, blocks);
        run_lds<1>("chain + ds_write_b64 per step", dOut, dCoef, blocks);
        run_lds<2>("chain + cvt + ds_write_b32 per step", dOut, dCoef, blocks);
        run_lds<3>("chain + cvt + ds_write_b16 per step", dOut, dCoef, blocks);
        run_lds<4>("chain + ds_write_b64 every 4th step", dOut, dCoef, blocks);
        run_lds<5>("chain + ds_write_b64 per step from 16 lanes", dOut, dCoef, blocks);
        run_lds<6>("chain + ds_write_b128 every 2nd step", dOut, dCoef, blocks);
        run_lds<7>("chain + ds_read_b64 per step", dOut, dCoef, blocks);
    }
    for (int blocks : {64}) {
        run_wg<0>("workgroup: nobody works (loop + barrier)", dOut, dCoef, blocks);
        run_wg<0, false>("workgroup: nobody works, no barrier", dOut, dCoef, blocks);
    }
    for (int blocks : {205}) {
        run_wg<1>("workgroup: source wave only", dOut, dCoef, blocks);
        run_wg<2>("workgroup: one filter wave only", dOut, dCoef, blocks);
        run_wg<6>("workgroup: both filter waves", dOut, dCoef, blocks);
        run_wg<8>("workgroup: final wave only", dOut, dCoef, blocks);
        run_wg<15>("workgroup: all four waves", dOut, dCoef, blocks);
        run_wg<2, false>("workgroup: one filter wave only, NO barrier (timing only)", dOut, dCoef, blocks);
        run_wg<15, false>("workgroup: all four waves, NO barrier (timing only)", dOut, dCoef, blocks);
        run_wg<2, false, 1>("one filter wave, no barrier, no loads", dOut, dCoef, blocks);
        run_wg<2, false, 2>("one filter wave, no barrier, no stores", dOut, dCoef, blocks);
        run_wg<2, false, 3>("one filter wave, no barrier, no loads, no stores", dOut, dCoef, blocks);
        run_wg<2, false, 7>("one filter wave, no barrier, no loads, stores, select", dOut, dCoef, blocks);
    }
    for (int blocks : {16, 205}) {
        // the filter step (variant 5) by how many waves of a 256-thread block run it, and which
        run_multi<5, 1, 1>("filter step, block of 1 wave", dOut, dCoef, blocks);
        run_multi<9, 1, 1>("filter step, input rows as in the kernel", dOut, dCoef, blocks);
        run_multi<10, 1, 1>("filter step, output rows as in the kernel", dOut, dCoef, blocks);
        run_multi<11, 1, 1>("filter step, both as in the kernel", dOut, dCoef, blocks);
        run_multi<5, 4, 1>("filter step, block of 4 waves, wave 0 works", dOut, dCoef, blocks);
        run_multi<5, 4, 2>("filter step, block of 4 waves, wave 1 works", dOut, dCoef, blocks);
        run_multi<5, 4, 3>("filter step, block of 4 waves, waves 0 and 1 work", dOut, dCoef, blocks);
        run_multi<5, 4, 15>("filter step, block of 4 waves, all four work", dOut, dCoef, blocks);
        run_multi<0, 4, 15>("resonator only, block of 4 waves, all four work", dOut, dCoef, blocks);
        run_multi<0, 1, 1>("resonator only, block of 1 wave", dOut, dCoef, blocks);
    }
    for (int blocks : {256, 2048}) {
        run<0>("resonator only (in = own output)", dOut, dCoef, blocks);
        run<1>("wave_ror:1 hand-over", dOut, dCoef, blocks);
        run<2>("row_shr:1 hand-over", dOut, dCoef, blocks);
        run<3>("ds_bpermute hand-over", dOut, dCoef, blocks);
        run<4>("wave_ror:1 + first-lane select", dOut, dCoef, blocks);
        run<5>("wave_ror:1 + select + LDS store", dOut, dCoef, blocks);
        run<6>("two interleaved pipelines (per step of both)", dOut, dCoef, blocks);
        run<7>("wave_ror:1 + select + store, FMA form (1 op on the chain)", dOut, dCoef, blocks);
        run<8>("row_shr:1 + select + LDS store", dOut, dCoef, blocks);
    }
    return 0;
}
