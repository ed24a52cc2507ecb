// locus_lik_common.hpp -- device helpers shared by the stage-1 likelihood kernels (locus_lik_kernel.hpp: eigenbasis value
// kernel + reverse-mode gradient; locus_value_kernel.hpp: value kernel on transition matrices).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

namespace tphip {

#ifndef TPHIP_LIK_BLOCK
#define TPHIP_LIK_BLOCK 128   // threads per workgroup of the locus likelihood kernels (a build-time knob for A/B runs)
#endif
constexpr int kLikBlock = TPHIP_LIK_BLOCK;
// A partial is rescaled when its largest entry falls below this.  Two partials meet in a POP_MUL before the next check,
// so the threshold must keep the PRODUCT of two just-unscaled partials above the smallest normal number (1e-308):
// 1e-200 let a 400-taxon tree underflow to L = 0.
constexpr double kLikRescaleBelow = 1e-100;
constexpr double kLikTiny = 1e-300;   // floor of a tip message entry (see the tip arm of the kernels)

// [lo, hi) of slice s of a locus spanning [llo, lhi)
__device__ inline void split_range(int64_t llo, int64_t lhi, int nsplit, int s, int block, int64_t* lo, int64_t* hi) {
    const int64_t len = lhi - llo;
    const int64_t per = ((len + nsplit - 1) / nsplit + block - 1) / block * block;
    const int64_t a = llo + (int64_t)s * per;
    *lo = a < lhi ? a : lhi;
    *hi = (a + per) < lhi ? (a + per) : lhi;
    if (*hi < *lo) *hi = *lo;
}

// Eigen-system of Q = R o pi for one candidate (single thread): eig = lam[4], U[16] (row-major, columns are right
// eigenvectors), Ui[16] = U^-1.  Symmetrised by sqrt(pi) and diagonalised with cyclic Jacobi sweeps.
__device__ inline void lik_eigen(const double* pi, const double* e, double* eig) {
    double R[4][4] = {{0, e[0], e[1], e[2]}, {e[0], 0, e[3], e[4]}, {e[1], e[3], 0, e[5]}, {e[2], e[4], e[5], 0}};
    double A[4][4], V[4][4], sq[4];
    for (int i = 0; i < 4; ++i) sq[i] = sqrt(pi[i]);
    for (int i = 0; i < 4; ++i) {
        double row = 0;
        for (int j = 0; j < 4; ++j)
            if (j != i) { row += R[i][j] * pi[j]; A[i][j] = sq[i] * R[i][j] * sq[j]; }
        A[i][i] = -row;
    }
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) V[i][j] = (i == j) ? 1.0 : 0.0;
    // cyclic Jacobi converges quadratically: once the off-diagonal mass is below eps^2 of the diagonal's, another sweep
    // changes nothing in double precision (the old absolute bound of 1e-290 cost three more sweeps out of nine)
    double diag2 = 0;
    for (int i = 0; i < 4; ++i) diag2 += A[i][i] * A[i][i];
    for (int sweep = 0; sweep < 30; ++sweep) {
        double off = 0;
        for (int p = 0; p < 4; ++p) for (int q = p + 1; q < 4; ++q) off += A[p][q] * A[p][q];
        if (off <= 1e-34 * diag2) break;
        for (int p = 0; p < 4; ++p) for (int q = p + 1; q < 4; ++q) {
            const double apq = A[p][q];
            if (apq == 0.0) continue;
            const double theta = (A[q][q] - A[p][p]) / (2.0 * apq);
            const double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
            const double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
            for (int k = 0; k < 4; ++k) { double x = A[k][p], y = A[k][q]; A[k][p] = c * x - s * y; A[k][q] = s * x + c * y; }
            for (int k = 0; k < 4; ++k) { double x = A[p][k], y = A[q][k]; A[p][k] = c * x - s * y; A[q][k] = s * x + c * y; }
            for (int k = 0; k < 4; ++k) { double x = V[k][p], y = V[k][q]; V[k][p] = c * x - s * y; V[k][q] = s * x + c * y; }
        }
    }
    for (int k = 0; k < 4; ++k) {
        eig[k] = A[k][k];
        for (int i = 0; i < 4; ++i) { eig[4 + i * 4 + k] = V[i][k] / sq[i]; eig[20 + k * 4 + i] = V[i][k] * sq[i]; }
    }
}

// wave-uniform double -> scalar registers
__device__ inline double lik_uniform(double v) {
    const unsigned long long b = __double_as_longlong(v);
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)b);
    const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(b >> 32));
    return __longlong_as_double(((unsigned long long)hi << 32) | lo);
}

}  // namespace tphip
