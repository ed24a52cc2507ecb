// mfma_f64_probe.hip -- what v_mfma_f64_4x4x4_4b_f64 is worth on gfx950 beside the FP64 VALU.
//
//   1. operand lane map: A one-hot per lane, B[lane] = lane + 1  ->  D tells which (A lane, B lane) pairs meet where;
//   2. issue cost in cycles per instruction (s_memtime around long unrolled loops, one wave per SIMD and two):
//      MFMA alone, FP64 FMA alone, and both interleaved as independent streams (same wave).
//
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/microbench/mfma_f64_probe tools/microbench/mfma_f64_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__global__ void layout_kernel(double* out) {  // out[64 a-lanes][64 lanes]
    const int lane = threadIdx.x;
    for (int la = 0; la < 64; ++la) {
        const double a = (lane == la) ? 1.0 : 0.0;
        const double b = (double)(lane + 1);
        const double d = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, 0.0, 0, 0, 0);
        out[la * 64 + lane] = d;
    }
}

template <int MODE>  // 0: MFMA only, 1: FMA only, 2: both (8 FMA per MFMA), 3: both (4 FMA per MFMA), 4: both (2 FMA per MFMA)
__global__ __launch_bounds__(512) void issue_kernel(double* sink, long long* cycles, int iters, double seed) {
    const int lane = threadIdx.x & 63;
    double a = seed + lane * 1e-3, b = 1.0 - lane * 1e-4;
    double d0 = 0, d1 = 0, d2 = 0, d3 = 0;
    double f0 = a, f1 = b, f2 = a + b, f3 = a - b, f4 = a * 0.5, f5 = b * 0.5, f6 = a * 0.25, f7 = b * 0.25;
    const double m = 0.9999999, c = 1e-9;
    const long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            if (MODE != 1) {
                d0 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, d0, 0, 0, 0);
                d1 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, d1, 0, 0, 0);
                d2 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, d2, 0, 0, 0);
                d3 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, d3, 0, 0, 0);
            }
            if (MODE == 1 || MODE == 2) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    f0 = fma(f0, m, c); f1 = fma(f1, m, c); f2 = fma(f2, m, c); f3 = fma(f3, m, c);
                    f4 = fma(f4, m, c); f5 = fma(f5, m, c); f6 = fma(f6, m, c); f7 = fma(f7, m, c);
                }
            }
            if (MODE == 3) {
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    f0 = fma(f0, m, c); f1 = fma(f1, m, c); f2 = fma(f2, m, c); f3 = fma(f3, m, c);
                    f4 = fma(f4, m, c); f5 = fma(f5, m, c); f6 = fma(f6, m, c); f7 = fma(f7, m, c);
                }
            }
            if (MODE == 4) {
                f0 = fma(f0, m, c); f1 = fma(f1, m, c); f2 = fma(f2, m, c); f3 = fma(f3, m, c);
                f4 = fma(f4, m, c); f5 = fma(f5, m, c); f6 = fma(f6, m, c); f7 = fma(f7, m, c);
            }
        }
    }
    const long long t1 = __builtin_readcyclecounter();
    sink[blockIdx.x * blockDim.x + threadIdx.x] = d0 + d1 + d2 + d3 + f0 + f1 + f2 + f3 + f4 + f5 + f6 + f7;
    if (lane == 0) cycles[blockIdx.x * (blockDim.x / 64) + (threadIdx.x >> 6)] = t1 - t0;
}

template <int MODE>
static void run_issue(const char* name, int threads, int iters, int n_mfma, int n_fma) {
    double* sink; long long* cyc;
    const int blocks = 256;
    CK(hipMalloc(&sink, sizeof(double) * blocks * threads));
    CK(hipMalloc(&cyc, sizeof(long long) * blocks * (threads / 64)));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    issue_kernel<MODE><<<blocks, threads>>>(sink, cyc, 10, 1.0);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    issue_kernel<MODE><<<blocks, threads>>>(sink, cyc, iters, 1.0);
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<long long> h(blocks * (threads / 64));
    CK(hipMemcpy(h.data(), cyc, sizeof(long long) * h.size(), hipMemcpyDeviceToHost));
    double avg = 0; for (auto v : h) avg += v; avg /= h.size();
    const double per_iter_ns = ms * 1e6 / iters;
    // per SIMD: waves/SIMD = threads/64/4 (one block per CU assumed: 256 blocks on 256 CUs)
    const int wps = threads / 256 ? threads / 256 : 1;
    printf("%-28s threads/block %4d (waves/SIMD %d): %8.3f ms, %7.1f ns per loop iteration per wave-set; per SIMD: "
           "%d MFMA + %d FMA per iteration -> %.2f ns per MFMA-equivalent slot [clock ticks/iter %.1f]\n",
           name, threads, wps, ms, per_iter_ns, n_mfma * wps, n_fma * wps, per_iter_ns / (wps * (n_mfma + n_fma / 4.0)), avg / iters);
    CK(hipFree(sink)); CK(hipFree(cyc));
}

int main() {
    double* out;
    CK(hipMalloc(&out, sizeof(double) * 64 * 64));
    layout_kernel<<<1, 64>>>(out);
    CK(hipDeviceSynchronize());
    std::vector<double> h(64 * 64);
    CK(hipMemcpy(h.data(), out, sizeof(double) * 64 * 64, hipMemcpyDeviceToHost));
    printf("# layout: A one-hot at lane la; B[lane]=lane+1; rows list 'dlane:value' for nonzero D\n");
    for (int la = 0; la < 64; ++la) {
        printf("la=%2d:", la);
        for (int l = 0; l < 64; ++l) if (h[la * 64 + l] != 0.0) printf(" %d:%g", l, h[la * 64 + l]);
        printf("\n");
    }
    const int iters = 20000;
    for (int threads : {256, 512}) {
        run_issue<0>("MFMA only (16/iter)", threads, iters, 16, 0);
        run_issue<1>("FMA only (128/iter)", threads, iters, 0, 128);
        run_issue<2>("MFMA 16 + FMA 128", threads, iters, 16, 128);
        run_issue<3>("MFMA 16 + FMA 64", threads, iters, 16, 64);
        run_issue<4>("MFMA 16 + FMA 32", threads, iters, 16, 32);
    }
    return 0;
}
