// locus_lik_kernel.hpp -- whole-locus log-likelihood for batches of candidate parameter sets.
//
// Foundation of HyPhy's stage 1 (tapir/data/models_and_rates.bf:405-897): every `Optimize(lf_MLES, lf)` there
// maximises sum over columns of log L(column | exchangeabilities, branch lengths) with all sites at rate 1
// (no per-site rate: that is stage 2).  The optimiser that drives it lives on the host (tapir_amd/stage1.py)
// and asks for MANY candidate points per locus at once (finite-difference gradients, line searches, the 203
// rate-class models), which is what makes the problem GPU-shaped: one workgroup per (locus, candidate).
//
//   1. thread 0 builds Q = R o pi for the candidate's exchangeabilities and diagonalises it (4x4 Jacobi);
//   2. all threads fill an LDS table exp(Lambda t_b), 4 doubles per tree branch;
//   3. threads stride over the slice's columns and prune with the same uniform op stream as site_rate_kernel,
//      value only, in the eigenbasis (see locus_loglik_kernel); parked siblings go to an LDS stack
//      [slot][state][thread];
//   4. log L summed over the columns by a fixed-order block reduction.
//
// Bound: FP64 VALU for short loci (the per-candidate eigen-system and ~130 matrix exponentials), the L2-resident
// byte reads of the alignment for long ones: ntaxa bytes and ~20 ntaxa flop per column per candidate.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include "gtr_model.hpp"
#include "tree_program.hpp"
#include "locus_lik_common.hpp"
#include "locus_lik_params.hpp"

namespace tphip {

// out[c][k] = sum_s part[c][s][k], s ascending
__global__ void split_sum_kernel(const double* part, double* out, int64_t ncand, int nsplit, int width) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= ncand * width) return;
    const int64_t c = i / width;
    const int k = (int)(i % width);
    double s = 0;
    for (int j = 0; j < nsplit; ++j) s += part[((size_t)c * nsplit + j) * width + k];
    out[i] = s;
}

// State masks of the block's columns: either staged in LDS (all loads of a column block in flight at once, then
// LDS-latency reads in the sweeps) or, for trees too large for that, read from global memory op by op.
__device__ inline void lik_stage_states(const LikParams& P, uint8_t* sts, int64_t c, int block) {
    for (int t = 0; t < P.ntaxa; ++t) {
        unsigned m = P.states[(int64_t)t * P.ncols_total + c] & 15u;
        sts[t * block + threadIdx.x] = (uint8_t)(m ? m : 15u);
    }
}
__device__ inline unsigned lik_tip_mask(const LikParams& P, const uint8_t* sts, int taxon, int64_t c, int block) {
    if (P.stage_states) return sts[taxon * block + threadIdx.x];
    const unsigned m = P.states[(int64_t)taxon * P.ncols_total + c] & 15u;
    return m ? m : 15u;
}

// locus_loglik_kernel: value only.  Works in the eigenbasis of Q = U Lambda U^-1 (reversible, so U^-1 = U^T diag(pi)):
//   tip    acc *= U (e_b o Y[mask])            Y[mask] = U^-1 (0/1 vector of the tip's state mask), 16-entry LDS table
//   branch acc  = U (e_b o U^T (pi o acc))     e_b = exp(Lambda t_b), 4 doubles per branch in LDS
// U, pi are wave-uniform and live in scalar registers; a tip costs 8 LDS reads and 24 FP64 instructions, no
// per-state select chains (the first version kept 4x4 transition matrices in LDS and was LDS-bandwidth bound).
__global__ __launch_bounds__(kLikBlock) void locus_loglik_kernel(LikParams P) {
    extern __shared__ double lds[];
    double* ET = lds;                                   // [nnodes][4] exp(lam_k t_b)
    double* stack = ET + (size_t)P.nnodes * 4;          // [depth][4][kLikBlock]
    uint8_t* sts = (uint8_t*)(stack + (size_t)P.stack_depth * 4 * kLikBlock);   // [ntaxa][kLikBlock] when staged
    __shared__ double eig[4 + 16 + 16];                 // lam[4], U[16], Ui[16]
    __shared__ double tipY[16 * 4];
    __shared__ double red[kLikBlock / 64];
    const int64_t cand = blockIdx.x / P.nsplit;
    const int locus = P.cand_locus[cand];
    int64_t lo, hi;
    split_range(P.locus_offsets[locus], P.locus_offsets[locus + 1], P.nsplit, (int)(blockIdx.x % P.nsplit), kLikBlock, &lo, &hi);
    if (lo >= hi) {   // empty slice (uniform for the block)
        if (threadIdx.x == 0) P.out[blockIdx.x] = 0.0;
        return;
    }
    const double* pig = P.models[locus].pi;
    if (threadIdx.x == 0) lik_eigen(pig, P.cand_exch + (size_t)cand * 6, eig);
    __syncthreads();
    const double* bl = P.blen_vecs + (size_t)P.cand_vec[cand] * P.nnodes;
    const double bscale = P.cand_scale[cand], pfac = P.cand_pfac[cand];
    const int pidx = P.cand_pidx[cand];
    for (int idx = threadIdx.x; idx < P.nnodes * 4; idx += kLikBlock) {
        const int b = idx >> 2, k = idx & 3;
        const double t = bl[b] * bscale * (b == pidx ? pfac : 1.0);
        ET[idx] = exp(eig[k] * t);
    }
    if (threadIdx.x < 64) {
        const int m = threadIdx.x >> 2, k = threadIdx.x & 3;
        double s = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) s += ((m >> j) & 1) ? eig[20 + k * 4 + j] : 0.0;
        tipY[threadIdx.x] = s;
    }
    __syncthreads();
    double U[16], pi[4];
#pragma unroll
    for (int i = 0; i < 16; ++i) U[i] = lik_uniform(eig[4 + i]);
#pragma unroll
    for (int k = 0; k < 4; ++k) pi[k] = lik_uniform(pig[k]);
    double total = 0.0;
    for (int64_t base = lo; base < hi; base += kLikBlock) {   // uniform trip count: every thread meets the LDS stack
        const int64_t col = base + threadIdx.x;
        const bool active = col < hi;
        const int64_t c = active ? col : lo;
        double acc[4] = {1.0, 1.0, 1.0, 1.0};
        int scale = 0, sp = 0;
        if (P.stage_states) lik_stage_states(P, sts, c, kLikBlock);
        int4 nxt = P.lops[0];
        for (int ip = 0; ip < P.nops; ++ip) {
            const int4 op = nxt;
            if (ip + 1 < P.nops) nxt = P.lops[ip + 1];
            if (op.x <= OP_TIP_MUL) {
                const unsigned m = lik_tip_mask(P, sts, op.y, c, kLikBlock);
                const double* et = ET + (size_t)op.z * 4;
                double z[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) z[k] = et[k] * tipY[m * 4 + k];
                // a message entry is a probability: for a near-zero branch the eigen-sum cancels to rounding noise around 0,
                // which must not turn the partial negative
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    acc[i] *= fmax(fma(U[i * 4 + 3], z[3], fma(U[i * 4 + 2], z[2], fma(U[i * 4 + 1], z[1], U[i * 4] * z[0]))), kLikTiny);
            } else if (op.x == OP_BRANCH) {
                const double mx = fmax(fmax(acc[0], acc[1]), fmax(acc[2], acc[3]));
                if (mx < kLikRescaleBelow && mx > 0) {   // rescale (deep trees); rare, lane-divergent is fine here
                    int e;
                    frexp(mx, &e);
                    for (int i = 0; i < 4; ++i) acc[i] = ldexp(acc[i], -e);
                    scale += e;
                }
                const double* et = ET + (size_t)op.z * 4;
                double w[4], y[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) w[i] = pi[i] * acc[i];
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    y[k] = et[k] * fma(U[12 + k], w[3], fma(U[8 + k], w[2], fma(U[4 + k], w[1], U[k] * w[0])));
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    acc[i] = fma(U[i * 4 + 3], y[3], fma(U[i * 4 + 2], y[2], fma(U[i * 4 + 1], y[1], U[i * 4] * y[0])));
            } else if (op.x == OP_PUSH) {
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
        if (active) {
            const double w = P.col_weight ? P.col_weight[c] : 1.0;
            total = fma(w, log(L) + (double)scale * 0.6931471805599453, total);
        }
    }
    // fixed-order block sum
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) total += __shfl_xor(total, o);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = total;
    __syncthreads();
    if (threadIdx.x == 0) {
        double s = 0;
        for (int w = 0; w < kLikBlock / 64; ++w) s += red[w];
        P.out[blockIdx.x] = s;
    }
}

// ---- value + gradient --------------------------------------------------------------------------------------
//
// locus_grad_kernel: log-likelihood of a candidate AND its derivatives w.r.t. every branch length and all six
// exchangeabilities from one forward + one reverse sweep over the op stream (reverse-mode differentiation of the
// pruning recursion), instead of 2 x (5 + 2N-3) extra likelihood evaluations for a finite-difference stencil.
//
// Everything is done in the eigenbasis of Q = U Lambda U^-1.  Q is reversible, so U^-1 = U^T diag(pi) and the
// three mat-vecs of a branch need U only (16 doubles, wave-uniform -> scalar registers) plus pi:
//       y = U^-1 a = U^T (pi o a),     message = U (e^{Lambda t} o y),     U^-T z = pi o (U z).
//
//   forward : tips  acc *= U (e_b o Y[mask])         (Y[mask] = U^-1 of the tip's 0/1 state vector, 16-entry LDS table)
//             inner acc  = U (e_b o U^T (pi o acc)); the partial entering every BRANCH and every PUSHed partial go
//             to a tape in global memory, [slot][state][thread] (coalesced; ~1.5 (N-2) slots of 32 B per column);
//   reverse : carries abar = d log L / d acc back through the ops.  For a branch b with incoming partial a and
//             outgoing adjoint abar, with x = U^T abar and y = U^-1 a:
//                 d log L / d t_b          = sum_k x_k lam_k e^{lam_k t_b} y_k
//                 d log L / d theta (in Q) = sum_kl x_k F_kl(t_b) y_l G_kl,   G = U^-1 (dQ/dtheta) U,
//                 F_kl = (e^{lam_k t} - e^{lam_l t}) / (lam_k - lam_l),  F_kk = t e^{lam_k t}   (Daleckii-Krein)
//             so the exchangeability derivatives of ALL branches and columns collapse into one 4x4 matrix
//             W_kl = sum x_k F_kl y_l per candidate, contracted with G for the six rates at the very end.
//             Tip branches are the same with y = Y[mask] and abar = the adjoint of the tip's message; the running
//             partial is un-multiplied (acc * 1/message).  Per-branch sums over the lanes go through LDS atomics
//             spread over kGradSlots addresses per (wave, branch).
//   outputs : lnL, d lnL / d r (6), d lnL / d log t_b (per branch, optional) and its sum over branches (what a
//             rate-class model needs, whose branch lengths are one stashed vector times 1/totalFactor(rates)).
//
// Bound: FP64 VALU (~150 flop-instructions per branch per column); the tape adds 64 B of HBM/L2 traffic per
// branch per column.  LDS holds only the per-branch tables (11 doubles) and the branch accumulators.
__device__ inline double lik_wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

__device__ inline double lik_rcp(double v) {   // 1/v to < 1 ulp-ish: hardware estimate + two Newton steps
    double r = __builtin_amdgcn_rcp(v);
    r = fma(fma(-v, r, 1.0), r, r);
    r = fma(fma(-v, r, 1.0), r, r);
    return r;
}

// The parameter block is passed by pointer: 25 pointers and sizes by value would sit in ~60 scalar registers for the
// whole kernel, next to the 48 of the eigenvectors, and spill into vector-register lanes inside the sweeps.
__global__ __launch_bounds__(kGradBlock, TPHIP_GRAD_MIN_WAVES) void locus_grad_kernel(const GradParams* __restrict__ Gp) {
    const GradParams& G = *Gp;
    const LikParams& P = G.L;
    extern __shared__ double lds[];
    const int nn = P.nnodes;
    double* EF = lds;                                   // [nn][kGradEF]
    const int ns = G.nslots;
    double* gb = EF + (size_t)nn * kGradEF;             // [kGradWaves][nn][ns]
    double* hb = gb + (size_t)kGradWaves * nn * ns;                 // same shape: second derivatives
    uint8_t* sts = (uint8_t*)(hb + (size_t)kGradWaves * nn * ns);   // [ntaxa][kGradBlock] when staged
    __shared__ double eig[36];
    __shared__ double tipY[16 * 4];
    __shared__ double red[kGradWaves * 18];
    const bool want_h = G.out_d2logt != nullptr;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    double* tape = G.tape + (size_t)blockIdx.x * (size_t)(G.ntape + P.stack_depth) * 4 * kGradBlock + tid;
    double* astack = tape + (size_t)G.ntape * 4 * kGradBlock;
    double* gbw = gb + ((size_t)wave * nn) * ns + (lane & (ns - 1));
    double* hbw = hb + ((size_t)wave * nn) * ns + (lane & (ns - 1));
    const int64_t nitems = G.ncand * P.nsplit;
    for (int64_t item = blockIdx.x; item < nitems; item += gridDim.x) {
        __syncthreads();
        const int64_t cand = item / P.nsplit;
        const int locus = P.cand_locus[cand];
        int64_t lo, hi;
        split_range(P.locus_offsets[locus], P.locus_offsets[locus + 1], P.nsplit, (int)(item % P.nsplit), kGradBlock, &lo, &hi);
        if (lo >= hi) {   // empty slice (uniform for the block): zero partials
            if (tid == 0) { P.out[item] = 0.0; G.out_sum_dlogt[item] = 0.0; }
            if (tid < 6) G.out_dexch[item * 6 + tid] = 0.0;
            if (G.out_dlogt) for (int b = tid; b < nn; b += kGradBlock) G.out_dlogt[item * nn + b] = 0.0;
            if (want_h) for (int b = tid; b < nn; b += kGradBlock) G.out_d2logt[item * nn + b] = 0.0;
            continue;
        }
        const double* pig = P.models[locus].pi;
        if (G.cand_eig) {
            if (tid < 36) eig[tid] = G.cand_eig[(size_t)cand * 36 + tid];
        } else if (tid == 0) {
            lik_eigen(pig, P.cand_exch + (size_t)cand * 6, eig);
        }
        __syncthreads();
        const double* bl = P.blen_vecs + (size_t)P.cand_vec[cand] * nn;
        const double bscale = P.cand_scale[cand], pfac = P.cand_pfac[cand];
        const int pidx = P.cand_pidx[cand];
        for (int b = tid; b < nn; b += kGradBlock) {
            const double t = bl[b] * bscale * (b == pidx ? pfac : 1.0);
            double e[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) { e[k] = exp(eig[k] * t); EF[b * kGradEF + k] = e[k]; }
            int q = 4;
#pragma unroll
            for (int k = 0; k < 4; ++k)
#pragma unroll
                for (int l = k + 1; l < 4; ++l) {
                    const double x = (eig[k] - eig[l]) * t;       // F_kl = t e^{lam_l t} expm1(x) / x
                    const double r = (fabs(x) < 1e-8) ? 1.0 + 0.5 * x : expm1(x) / x;
                    EF[b * kGradEF + q++] = t * e[l] * r;
                }
            EF[b * kGradEF + 10] = t;
            EF[b * kGradEF + 11] = 0.0;
            for (int w = 0; w < kGradWaves * ns; ++w) {
                gb[((size_t)(w / ns) * nn + b) * ns + (w % ns)] = 0.0;
                hb[((size_t)(w / ns) * nn + b) * ns + (w % ns)] = 0.0;
            }
        }
        if (tid < 64) {  // Y[mask] = U^-1 (0/1 vector of the state mask)
            const int m = tid >> 2, k = tid & 3;
            double s = 0;
#pragma unroll
            for (int j = 0; j < 4; ++j) s += ((m >> j) & 1) ? eig[20 + k * 4 + j] : 0.0;
            tipY[tid] = s;
        }
        __syncthreads();
        double U[16], lam[4], pi[4];
#pragma unroll
        for (int i = 0; i < 16; ++i) U[i] = lik_uniform(eig[4 + i]);
#pragma unroll
        for (int k = 0; k < 4; ++k) { lam[k] = lik_uniform(eig[k]); pi[k] = lik_uniform(pig[k]); }
        double total = 0.0;
        double inv_seedw = 0.0;   // 1 / column weight of the column in flight (0 for padding lanes)
        double W[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) W[i] = 0.0;

        // r = U z   and   r = U^T z
        auto mulU = [&](const double* z, double* r) {
#pragma unroll
            for (int i = 0; i < 4; ++i) r[i] = fma(U[i * 4 + 3], z[3], fma(U[i * 4 + 2], z[2], fma(U[i * 4 + 1], z[1], U[i * 4] * z[0])));
        };
        auto mulUt = [&](const double* z, double* r) {
#pragma unroll
            for (int k = 0; k < 4; ++k) r[k] = fma(U[12 + k], z[3], fma(U[8 + k], z[2], fma(U[4 + k], z[1], U[k] * z[0])));
        };
        // x = U^T abar, y = U^-1 a: accumulate W and the branch-length derivative of `node`
        auto contribute = [&](const double* ef, int node, const double* x, const double* y) {
            const double e0 = ef[0], e1 = ef[1], e2 = ef[2], e3 = ef[3], t = ef[10];
            const double f01 = ef[4], f02 = ef[5], f03 = ef[6], f12 = ef[7], f13 = ef[8], f23 = ef[9];
            const double d0 = x[0] * y[0] * e0, d1 = x[1] * y[1] * e1, d2 = x[2] * y[2] * e2, d3 = x[3] * y[3] * e3;
            W[0] = fma(t, d0, W[0]);                  W[1] = fma(x[0] * f01, y[1], W[1]);
            W[2] = fma(x[0] * f02, y[2], W[2]);       W[3] = fma(x[0] * f03, y[3], W[3]);
            W[4] = fma(x[1] * f01, y[0], W[4]);       W[5] = fma(t, d1, W[5]);
            W[6] = fma(x[1] * f12, y[2], W[6]);       W[7] = fma(x[1] * f13, y[3], W[7]);
            W[8] = fma(x[2] * f02, y[0], W[8]);       W[9] = fma(x[2] * f12, y[1], W[9]);
            W[10] = fma(t, d2, W[10]);                W[11] = fma(x[2] * f23, y[3], W[11]);
            W[12] = fma(x[3] * f03, y[0], W[12]);     W[13] = fma(x[3] * f13, y[1], W[13]);
            W[14] = fma(x[3] * f23, y[2], W[14]);     W[15] = fma(t, d3, W[15]);
            const double c = fma(lam[3], d3, fma(lam[2], d2, fma(lam[1], d1, lam[0] * d0)));
            atomicAdd(gbw + (size_t)node * ns, c);
            if (want_h) {   // d2 log L / dt^2 of this column = L''/L - (L'/L)^2; the adjoint already carries weight / L
                const double c2 = fma(lam[3] * lam[3], d3, fma(lam[2] * lam[2], d2, fma(lam[1] * lam[1], d1, lam[0] * lam[0] * d0)));
                atomicAdd(hbw + (size_t)node * ns, c2 - c * c * inv_seedw);
            }
        };

        for (int64_t base = lo; base < hi; base += kGradBlock) {
            const int64_t col = base + tid;
            const bool active = col < hi;
            const int64_t c = active ? col : lo;
            double acc[4] = {1.0, 1.0, 1.0, 1.0};
            int scale = 0;
            if (P.stage_states) lik_stage_states(P, sts, c, kGradBlock);
            // ---------------- forward ----------------
            int4 nxt = P.lops[0];
            for (int ip = 0; ip < P.nops; ++ip) {
                const int4 op = nxt;
                if (ip + 1 < P.nops) nxt = P.lops[ip + 1];
                if (op.x <= OP_TIP_MUL) {
                    const unsigned m = lik_tip_mask(P, sts, op.y, c, kGradBlock);
                    const double* ef = EF + (size_t)op.z * kGradEF;
                    double z[4], v[4];
#pragma unroll
                    for (int k = 0; k < 4; ++k) z[k] = ef[k] * tipY[m * 4 + k];
                    mulU(z, v);
#pragma unroll
                    for (int i = 0; i < 4; ++i) acc[i] *= fmax(v[i], kLikTiny);
                } else if (op.x == OP_BRANCH) {
                    double* slot = tape + (size_t)op.w * 4 * kGradBlock;
#pragma unroll
                    for (int i = 0; i < 4; ++i) slot[i * kGradBlock] = acc[i];   // raw: the reverse sweep redoes the rescale
                    const double mx = fmax(fmax(acc[0], acc[1]), fmax(acc[2], acc[3]));
                    if (mx < kLikRescaleBelow && mx > 0) {
                        int e;
                        frexp(mx, &e);
                        for (int i = 0; i < 4; ++i) acc[i] = ldexp(acc[i], -e);
                        scale += e;
                    }
                    const double* ef = EF + (size_t)op.z * kGradEF;
                    double w[4], y[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) w[i] = pi[i] * acc[i];
                    mulUt(w, y);
#pragma unroll
                    for (int k = 0; k < 4; ++k) y[k] *= ef[k];
                    mulU(y, acc);
                } else if (op.x == OP_PUSH) {
                    double* slot = tape + (size_t)op.w * 4 * kGradBlock;
#pragma unroll
                    for (int i = 0; i < 4; ++i) { slot[i * kGradBlock] = acc[i]; acc[i] = 1.0; }
                } else {
                    const double* slot = tape + (size_t)op.w * 4 * kGradBlock;
#pragma unroll
                    for (int i = 0; i < 4; ++i) acc[i] *= slot[i * kGradBlock];
                }
            }
            const double Lc = fma(pi[3], acc[3], fma(pi[2], acc[2], fma(pi[1], acc[1], pi[0] * acc[0])));
            const double cw = active ? (P.col_weight ? P.col_weight[c] : 1.0) : 0.0;
            total = fma(cw, log(Lc) + (double)scale * 0.6931471805599453, total);
            // ---------------- reverse ----------------
            const double seed = cw / Lc;   // padding lanes contribute exact zeros
            inv_seedw = cw > 0.0 ? 1.0 / cw : 0.0;
            double ab[4] = {pi[0] * seed, pi[1] * seed, pi[2] * seed, pi[3] * seed};
            int asp = 0;
            nxt = P.lops[P.nops - 1];
            for (int ip = P.nops - 1; ip >= 0; --ip) {
                const int4 op = nxt;
                if (ip > 0) nxt = P.lops[ip - 1];
                if (op.x <= OP_TIP_MUL) {
                    const unsigned m = lik_tip_mask(P, sts, op.y, c, kGradBlock);
                    const int node = op.z;
                    const double* ef = EF + (size_t)node * kGradEF;
                    double y[4], z[4], v[4], vb[4], x[4];
#pragma unroll
                    for (int k = 0; k < 4; ++k) { y[k] = tipY[m * 4 + k]; z[k] = ef[k] * y[k]; }
                    mulU(z, v);
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        v[i] = fmax(v[i], kLikTiny);
                        acc[i] *= lik_rcp(v[i]);    // the partial before this tip was folded in
                        vb[i] = ab[i] * acc[i];     // adjoint of the tip's message
                        ab[i] *= v[i];
                    }
                    mulUt(vb, x);
                    contribute(ef, node, x, y);
                } else if (op.x == OP_BRANCH) {
                    const int node = op.z;
                    const double* slot = tape + (size_t)op.w * 4 * kGradBlock;
                    double a[4], as[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) { a[i] = slot[i * kGradBlock]; as[i] = a[i]; }
                    double sc = 1.0;
                    const double mx = fmax(fmax(a[0], a[1]), fmax(a[2], a[3]));
                    if (mx < kLikRescaleBelow && mx > 0) {
                        int e;
                        frexp(mx, &e);
                        for (int i = 0; i < 4; ++i) as[i] = ldexp(a[i], -e);
                        sc = ldexp(1.0, -e);
                    }
                    const double* ef = EF + (size_t)node * kGradEF;
                    double w[4], x[4], y[4], z[4], r[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) w[i] = pi[i] * as[i];
                    mulUt(w, y);
                    mulUt(ab, x);
                    contribute(ef, node, x, y);
#pragma unroll
                    for (int k = 0; k < 4; ++k) z[k] = ef[k] * x[k];
                    mulU(z, r);
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        ab[j] = sc * pi[j] * r[j];
                        acc[j] = a[j];
                    }
                } else if (op.x == OP_PUSH) {
                    --asp;
                    const double* as_ = astack + ((size_t)asp * 4) * kGradBlock;
                    const double* slot = tape + (size_t)op.w * 4 * kGradBlock;
#pragma unroll
                    for (int i = 0; i < 4; ++i) { ab[i] = as_[i * kGradBlock]; acc[i] = slot[i * kGradBlock]; }
                } else {
                    const double* slot = tape + (size_t)op.w * 4 * kGradBlock;
                    double* as_ = astack + ((size_t)asp * 4) * kGradBlock;
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const double S = slot[i * kGradBlock];
                        acc[i] *= lik_rcp(S);                // the partial that was multiplied by the parked one
                        as_[i * kGradBlock] = ab[i] * acc[i];
                        ab[i] *= S;
                    }
                    ++asp;
                }
            }
        }
        // ---------------- reductions (fixed order) ----------------
        total = lik_wave_sum(total);
#pragma unroll
        for (int i = 0; i < 16; ++i) W[i] = lik_wave_sum(W[i]);
        if (lane == 0) {
            red[wave * 18] = total;
            for (int i = 0; i < 16; ++i) red[wave * 18 + 1 + i] = W[i];
        }
        __syncthreads();
        if (tid == 0) {
            double lnl = 0, Wt[16];
            for (int i = 0; i < 16; ++i) Wt[i] = 0;
            for (int w = 0; w < kGradWaves; ++w) {
                lnl += red[w * 18];
                for (int i = 0; i < 16; ++i) Wt[i] += red[w * 18 + 1 + i];
            }
            P.out[item] = lnl;
            // dQ/dr_ij = pi_j (E_ij - E_ii) + pi_i (E_ji - E_jj)  ->  G_kl = (U_jl - U_il) (pi_j Ui_ki - pi_i Ui_kj)
            const int pi_[6] = {0, 0, 0, 1, 1, 2}, pj_[6] = {1, 2, 3, 2, 3, 3};
            for (int q = 0; q < 6; ++q) {
                const int i = pi_[q], j = pj_[q];
                double d = 0;
                for (int k = 0; k < 4; ++k)
                    for (int l = 0; l < 4; ++l)
                        d += Wt[k * 4 + l] * (eig[4 + j * 4 + l] - eig[4 + i * 4 + l]) * (pig[j] * eig[20 + k * 4 + i] - pig[i] * eig[20 + k * 4 + j]);
                G.out_dexch[item * 6 + q] = d;
            }
        }
        // d lnL / d log t_b = t_b * sum over waves and slots; and its total
        double part = 0.0;
        for (int b = tid; b < nn; b += kGradBlock) {
            double g = 0;
            for (int w = 0; w < kGradWaves; ++w)
                for (int s = 0; s < ns; ++s) g += gb[((size_t)w * nn + b) * ns + s];
            const double t = EF[b * kGradEF + 10];
            if (want_h) {   // d2/d(log t)^2 = t^2 d2/dt^2 + t d/dt
                double h = 0;
                for (int w = 0; w < kGradWaves; ++w)
                    for (int s = 0; s < ns; ++s) h += hb[((size_t)w * nn + b) * ns + s];
                G.out_d2logt[item * nn + b] = t * (t * h + g);
            }
            g *= t;
            if (G.out_dlogt) G.out_dlogt[item * nn + b] = g;
            part += g;
        }
        part = lik_wave_sum(part);
        __syncthreads();   // red[] was read by thread 0 above
        if (lane == 0) red[wave] = part;
        __syncthreads();
        if (tid == 0) {
            double s = 0;
            for (int w = 0; w < kGradWaves; ++w) s += red[w];
            G.out_sum_dlogt[item] = s;
        }
    }
}

}  // namespace tphip
