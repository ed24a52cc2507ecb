// locus_lik_kernel.hpp -- whole-locus log-likelihood for batches of candidate parameter sets.
//
// Foundation of HyPhy's stage 1 (tapir/data/models_and_rates.bf:405-897): every `Optimize(lf_MLES, lf)` there
// maximises sum over columns of log L(column | exchangeabilities, branch lengths) with all sites at rate 1
// (no per-site rate: that is stage 2).  The optimiser that drives it lives on the host (tapir_amd/stage1.py)
// and asks for MANY candidate points per locus at once (finite-difference gradients, line searches, the 203
// rate-class models), which is what makes the problem GPU-shaped: one workgroup per (locus, candidate).
//
//   1. thread 0 builds Q = R o pi for the candidate's exchangeabilities and diagonalises it (4x4 Jacobi);
//   2. all threads fill an LDS table of transition matrices P_b = U exp(Lambda t_b) U^-1, one per tree branch;
//   3. threads stride over the locus' columns and prune with the same uniform op stream as site_rate_kernel,
//      value only: a tip contributes sum_{x in mask} P_b[.][x], an internal branch one 4x4 mat-vec; parked
//      siblings go to an LDS stack [slot][state][thread];
//   4. log L summed over the columns by a fixed-order block reduction.
//
// Bound: FP64 VALU for short loci (the per-candidate eigen-system and ~130 matrix exponentials), the L2-resident
// byte reads of the alignment for long ones: ntaxa bytes and ~20 ntaxa flop per column per candidate.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include "gtr_model.hpp"
#include "tree_program.hpp"

namespace tphip {

constexpr int kLikBlock = 256;

struct LikParams {
    const uint8_t* states;         // [ntaxa][ncols_total]
    int64_t ncols_total;
    const int64_t* locus_offsets;  // [nloci+1]
    const LocusModel* models;      // [nloci] (only pi is used)
    const TreeOp* ops;             // shared traversal program
    const int32_t* op_node;        // [nops] node whose branch the op climbs
    int32_t nops;
    int32_t nnodes;
    int32_t stack_depth;
    const int32_t* cand_locus;     // [ncand]
    const double* cand_exch;       // [ncand][6] AC,AG,AT,CG,CT,GT
    // branch lengths of candidate c: blen_vecs[cand_vec[c]][b] * cand_scale[c] * (b == cand_pidx[c] ? cand_pfac[c] : 1).
    // A finite-difference stencil around one point, or the 202 rate-class models of one locus (which only rescale
    // the stashed lengths, bf:613-619), therefore share ONE stored vector instead of carrying nnodes doubles each.
    const double* blen_vecs;       // [nvec][nnodes]
    const int32_t* cand_vec;       // [ncand]
    const double* cand_scale;      // [ncand]
    const int32_t* cand_pidx;      // [ncand] node whose branch is perturbed, -1 for none
    const double* cand_pfac;       // [ncand]
    double* out;                   // [ncand] sum over columns of log L
};

__global__ __launch_bounds__(kLikBlock) void locus_loglik_kernel(LikParams P) {
    extern __shared__ double lds[];
    double* Pm = lds;                                   // [nnodes][16]
    double* stack = Pm + (size_t)P.nnodes * 16;         // [depth][4][kLikBlock]
    __shared__ double eig[4 + 16 + 16];                 // lam[4], U[16], Ui[16]
    __shared__ double red[kLikBlock / 64];
    const int cand = blockIdx.x;
    const int locus = P.cand_locus[cand];
    const double* pi = P.models[locus].pi;
    if (threadIdx.x == 0) {
        const double* e = P.cand_exch + (size_t)cand * 6;
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
        for (int sweep = 0; sweep < 30; ++sweep) {
            double off = 0;
            for (int p = 0; p < 4; ++p) for (int q = p + 1; q < 4; ++q) off += A[p][q] * A[p][q];
            if (off < 1e-290) break;
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
    __syncthreads();
    // transition matrices: P_b[i][j] = sum_k U[i][k] exp(lam_k t_b) Ui[k][j]
    const double* bl = P.blen_vecs + (size_t)P.cand_vec[cand] * P.nnodes;
    const double bscale = P.cand_scale[cand], pfac = P.cand_pfac[cand];
    const int pidx = P.cand_pidx[cand];
    for (int idx = threadIdx.x; idx < P.nnodes * 16; idx += kLikBlock) {
        const int b = idx >> 4, i = (idx >> 2) & 3, j = idx & 3;
        const double t = bl[b] * bscale * (b == pidx ? pfac : 1.0);
        double s = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) s = fma(eig[4 + i * 4 + k] * exp(eig[k] * t), eig[20 + k * 4 + j], s);
        Pm[idx] = s;
    }
    __syncthreads();
    const int64_t lo = P.locus_offsets[locus], hi = P.locus_offsets[locus + 1];
    double total = 0.0;
    for (int64_t base = lo; base < hi; base += kLikBlock) {   // uniform trip count: every thread meets the LDS stack
        const int64_t col = base + threadIdx.x;
        const bool active = col < hi;
        const int64_t c = active ? col : lo;
        double acc[4] = {1.0, 1.0, 1.0, 1.0};
        int scale = 0, sp = 0;
        for (int ip = 0; ip < P.nops; ++ip) {
            const TreeOp op = P.ops[ip];
            if (op.code <= OP_TIP_MUL) {
                unsigned m = P.states[(int64_t)op.taxon * P.ncols_total + c] & 15u;
                m = m ? m : 15u;
                const double* Pb = Pm + (size_t)P.op_node[ip] * 16;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    double v = 0;
#pragma unroll
                    for (int j = 0; j < 4; ++j) v += ((m >> j) & 1u) ? Pb[i * 4 + j] : 0.0;
                    acc[i] *= v;
                }
            } else if (op.code == OP_BRANCH) {
                const double mx = fmax(fmax(acc[0], acc[1]), fmax(acc[2], acc[3]));
                if (mx < 1e-200 && mx > 0) {   // rescale (deep trees); rare, lane-divergent is fine here
                    int e;
                    frexp(mx, &e);
                    for (int i = 0; i < 4; ++i) acc[i] = ldexp(acc[i], -e);
                    scale += e;
                }
                const double* Pb = Pm + (size_t)P.op_node[ip] * 16;
                double r[4];
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    r[i] = fma(Pb[i * 4 + 3], acc[3], fma(Pb[i * 4 + 2], acc[2], fma(Pb[i * 4 + 1], acc[1], Pb[i * 4] * acc[0])));
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[i] = r[i];
            } else if (op.code == OP_PUSH) {
                double* slot = stack + ((size_t)sp * 4) * kLikBlock + threadIdx.x;
#pragma unroll
                for (int i = 0; i < 4; ++i) { slot[i * kLikBlock] = acc[i]; acc[i] = 1.0; }
                ++sp;
            } else {
                --sp;
                const double* slot = stack + ((size_t)sp * 4) * kLikBlock + threadIdx.x;
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[i] *= slot[i * kLikBlock];
            }
        }
        const double L = fma(pi[3], acc[3], fma(pi[2], acc[2], fma(pi[1], acc[1], pi[0] * acc[0])));
        if (active) total += log(L) + (double)scale * 0.6931471805599453;
    }
    // fixed-order block sum
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) total += __shfl_xor(total, o);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = total;
    __syncthreads();
    if (threadIdx.x == 0) P.out[cand] = ((red[0] + red[1]) + red[2]) + red[3];
}

}  // namespace tphip
