// gtr_setup_kernel.hpp -- per-locus GTR eigen-systems on the device (see gtr_model.hpp for the model and the layout).
// Included by tphip.hip only (a __global__ definition must live in one translation unit).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include "gtr_model.hpp"

namespace tphip {

__global__ void gtr_setup_kernel(const double* __restrict__ pi_in, const double* __restrict__ exch_in, int64_t nloci,
                                 LocusModel* __restrict__ models) {
    int64_t l = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (l >= nloci) return;
    double pi[4], sq[4];
    double psum = 0;
    for (int i = 0; i < 4; ++i) { pi[i] = pi_in[l * 4 + i]; psum += pi[i]; }
    for (int i = 0; i < 4; ++i) { pi[i] /= psum; sq[i] = sqrt(pi[i]); }
    const double* e = exch_in + l * 6;  // AC AG AT CG CT GT
    double R[4][4] = {{0, e[0], e[1], e[2]}, {e[0], 0, e[3], e[4]}, {e[1], e[3], 0, e[5]}, {e[2], e[4], e[5], 0}};
    double A[4][4], V[4][4];
    double kappa = 0;
    for (int i = 0; i < 4; ++i) {
        double row = 0;
        for (int j = 0; j < 4; ++j)
            if (j != i) { row += R[i][j] * pi[j]; A[i][j] = sq[i] * R[i][j] * sq[j]; }  // S_ij = sqrt(pi_i) r_ij sqrt(pi_j)
        A[i][i] = -row;
        kappa += pi[i] * row;
    }
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) V[i][j] = (i == j) ? 1.0 : 0.0;
    for (int sweep = 0; sweep < 30; ++sweep) {
        double off = 0;
        for (int p = 0; p < 4; ++p) for (int q = p + 1; q < 4; ++q) off += A[p][q] * A[p][q];
        if (off < 1e-290) break;
        for (int p = 0; p < 4; ++p) for (int q = p + 1; q < 4; ++q) {
            double apq = A[p][q];
            if (apq == 0.0) continue;
            double theta = (A[q][q] - A[p][p]) / (2.0 * apq);
            double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
            double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
            for (int k = 0; k < 4; ++k) { double x = A[k][p], y = A[k][q]; A[k][p] = c * x - s * y; A[k][q] = s * x + c * y; }
            for (int k = 0; k < 4; ++k) { double x = A[p][k], y = A[q][k]; A[p][k] = c * x - s * y; A[q][k] = s * x + c * y; }
            for (int k = 0; k < 4; ++k) { double x = V[k][p], y = V[k][q]; V[k][p] = c * x - s * y; V[k][q] = s * x + c * y; }
        }
    }
    // slot 0 <- the eigenvalue closest to zero (the largest: all others are negative)
    int z = 0;
    for (int k = 1; k < 4; ++k) if (A[k][k] > A[z][z]) z = k;
    int order[4] = {z, 0, 0, 0};
    for (int k = 0, o = 1; k < 4; ++k) if (k != z) order[o++] = k;
    LocusModel m;
    for (int o = 1; o < 4; ++o) {
        int k = order[o];
        m.lam[o - 1] = A[k][k];
        for (int i = 0; i < 4; ++i) {
            m.U[i * 3 + (o - 1)] = V[i][k] / sq[i];
            m.Ui[(o - 1) * 4 + i] = V[i][k] * sq[i];
        }
    }
    for (int i = 0; i < 4; ++i) m.pi[i] = pi[i];
    m.kappa = kappa;
    models[l] = m;
}

}  // namespace tphip
