// fp64_operand_probe.hip -- does the source-operand mix change the issue rate of FP64 VALU instructions on gfx950?
// Eight independent chains per wave, 2 waves per SIMD (512 threads per CU), 256 CUs.  Variants:
//   vss  v_fma_f64 d, d, s, s      one VGPR source (the chain), two SGPR sources
//   vvs  v_fma_f64 d, v, s, d      two VGPR sources + one SGPR
//   vvv  v_fma_f64 d, v, v, d      three VGPR sources (what site_rate_kernel issues: model in VGPRs)
//   fmac v_fmac_f64 d, v, v        VOP2 form, three VGPR reads
//   mul  v_mul_f64 d, d, v
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/microbench/fp64_operand_probe tools/microbench/fp64_operand_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

#define CHAIN8(STMT) STMT(f0) STMT(f1) STMT(f2) STMT(f3) STMT(f4) STMT(f5) STMT(f6) STMT(f7)
#define VSS(f) asm volatile("v_fma_f64 %0, %0, %1, 0.5" : "+v"(f) : "s"(ms));
#define VVS(f) asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(f) : "v"(a), "s"(ms));
#define VVV(f) asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(f) : "v"(a), "v"(b));
#define FMAC(f) asm volatile("v_fmac_f64 %0, %1, %2" : "+v"(f) : "v"(a), "v"(b));
#define MUL(f) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(f) : "v"(a));
#define VVV2(f) asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(f) : "v"(a2), "v"(b2));

template <int MODE>
__global__ __launch_bounds__(512) void k(double* sink, int iters, double ms, double cs) {
    const int lane = threadIdx.x & 63;
    double a = 1e-9 * lane, b = 1.0 - 1e-7 * lane, a2 = 1e-8 * lane, b2 = 0.5 + 1e-7 * lane;
    double f0 = a, f1 = b, f2 = a + b, f3 = a - b, f4 = a * 0.5, f5 = b * 0.5, f6 = a * 0.25, f7 = b * 0.25;
    asm volatile("" : "+v"(a), "+v"(b), "+v"(a2), "+v"(b2));
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            if (MODE == 0) { CHAIN8(VSS) }
            if (MODE == 1) { CHAIN8(VVS) }
            if (MODE == 2) { CHAIN8(VVV) }
            if (MODE == 3) { CHAIN8(FMAC) }
            if (MODE == 4) { CHAIN8(MUL) }
            if (MODE == 5) { VVV(f0) VVV2(f1) VVV(f2) VVV2(f3) VVV(f4) VVV2(f5) VVV(f6) VVV2(f7) }
        }
    }
    sink[blockIdx.x * blockDim.x + threadIdx.x] = f0 + f1 + f2 + f3 + f4 + f5 + f6 + f7;
}

template <int MODE>
static void run(const char* name) {
    double* sink;
    const int blocks = 256, threads = 512, iters = 20000;
    CK(hipMalloc(&sink, sizeof(double) * blocks * threads));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    k<MODE><<<blocks, threads>>>(sink, 10, 0.9999999, 1e-9);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    k<MODE><<<blocks, threads>>>(sink, iters, 0.9999999, 1e-9);
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    float msec; CK(hipEventElapsedTime(&msec, e0, e1));
    const double per_simd = 2.0 * 64 * iters;   // 2 waves x 64 instructions per iteration
    printf("%-6s %8.3f ms  %.3f ns per wave-instruction per SIMD\n", name, msec, msec * 1e6 / per_simd);
    CK(hipFree(sink));
}

int main() {
    run<0>("vss"); run<1>("vvs"); run<2>("vvv"); run<3>("fmac"); run<4>("mul"); run<5>("vvv-2");
    return 0;
}
