// locus_grad2_kernel.hpp -- whole-locus log-likelihood AND its gradient from per-branch transition matrices.
//
// What it computes is what locus_grad_kernel (locus_lik_kernel.hpp) computes -- for a batch of candidate parameter sets of
// HyPhy's stage-1 objective (tapir/data/models_and_rates.bf:487-520: Optimize over the exchangeabilities and every branch
// length), per candidate: lnL, d lnL / d (AC, AG, AT, CG, CT, GT) at fixed branch lengths, d lnL / d log t_b and
// d2 lnL / d (log t_b)^2 for every branch -- by one forward and one reverse sweep of the pruning recursion.  How differs:
//
//   forward   = locus_value_kernel's sweep (locus_value_kernel.hpp): tips are rows of an LDS copy of the candidate's tip
//               matrices, an internal branch is acc = P_b acc with P_b in scalar registers (16 FP64 instructions; the
//               eigenbasis form of locus_grad_kernel takes 40), parked siblings in a register stack, fused op stream.  The
//               MESSAGE P_b acc of every internal branch goes to a tape in global memory, [slot][state][thread]: one 32-byte
//               slot per internal branch and column, and nothing else -- locus_grad_kernel tapes ~1.5 slots per branch plus
//               an adjoint stack, and recovers the running products by dividing them out again (four reciprocals per op).
//   reverse   = a pre-order walk over the INTERNAL NODES (a program of its own, locus_grad2_params.hpp).  At node n with
//               children A, B and the adjoint u_n of its message: l_n = m_A o m_B from the tape (tips: the LDS rows again),
//               x = U^T u_n, y = U^-1 l_n = U^T (pi o l_n); the branch's contribution to every derivative is
//                   d lnL / d t_n          = sum_k x_k lam_k e^{lam_k t} y_k
//                   d lnL / d theta (in Q) = sum_kl x_k F_kl(t_n) y_l G_kl      (Daleckii-Krein, G = U^-1 (dQ/dtheta) U)
//               with G symmetric in the eigenbasis of a reversible Q, so only x_k y_l + x_l y_k is accumulated: 10 running
//               sums per lane instead of 16.  Then a_n = pi o U (e o x) is the adjoint of l_n, u_A = a_n o m_B, u_B = a_n o m_A,
//               and tip children contribute at once with y = U^-1 (state vector) from a 16-entry LDS table.  An internal
//               child's adjoint is carried in registers to the next record; when both are internal the second is parked in
//               LDS.  No division anywhere.
//   per-branch sums over the lanes: row sums by data-parallel primitives (no LDS round trip), then the four row leaders of a
//   wave add to the wave's row of LDS accumulators.
//
// Binary trees only (every internal node has two children) and the packed state codes of the value kernel; anything else
// runs locus_grad_kernel.  Bound: FP64 VALU, ~150 instructions per internal branch and ~75 per tip branch per column
// (locus_grad_kernel: ~150 for both, plus the reciprocals), and 64 B of tape traffic per internal branch and column.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include <type_traits>
#include "gtr_model.hpp"
#include "tree_program.hpp"
#include "locus_lik_common.hpp"
#include "locus_value_params.hpp"
#include "locus_grad2_params.hpp"

namespace tphip {

typedef const double __attribute__((address_space(4)))* value_cptr;   // constant address space: wave-uniform reads become scalar loads
typedef int g2_i4 __attribute__((ext_vector_type(4)));               // an op record as a native vector (loadable from address space 4)
typedef const g2_i4 __attribute__((address_space(4)))* g2_ops_cptr;

// ef[cand][node] = e^{lam_k t}[4], F_kl(t) for k < l [6], t, 0   (F_kl = (e^{lam_k t} - e^{lam_l t}) / (lam_k - lam_l))
__global__ __launch_bounds__(128) void grad2_ef_kernel(const double* eig, const double* blen_vecs, const int32_t* cand_vec,
                                                       const double* cand_scale, const int32_t* cand_pidx, const double* cand_pfac,
                                                       int64_t ncand, int32_t nnodes, double* ef) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= ncand * nnodes) return;
    const int64_t c = idx / nnodes;
    const int b = (int)(idx - c * nnodes);
    const double* E = eig + (size_t)c * 36;
    const double t = blen_vecs[(size_t)cand_vec[c] * nnodes + b] * cand_scale[c] * (b == cand_pidx[c] ? cand_pfac[c] : 1.0);
    double e[4];
    double* out = ef + (size_t)idx * kGrad2EF;
#pragma unroll
    for (int k = 0; k < 4; ++k) { e[k] = exp(E[k] * t); out[k] = e[k]; }
    int q = 4;
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int l = k + 1; l < 4; ++l) {
            const double x = (E[k] - E[l]) * t;       // F_kl = t e^{lam_l t} expm1(x) / x
            const double r = (fabs(x) < 1e-8) ? 1.0 + 0.5 * x : expm1(x) / x;
            out[q++] = t * e[l] * r;
        }
    out[10] = t;
    out[11] = 0.0;
}

typedef double __attribute__((address_space(1)))* g2_gptr;             // global address space: plain global_load / global_store, not flat
typedef const uint32_t __attribute__((address_space(1)))* g2_gcu32;

// v of lane i += v of lane i - n within its row of 16 lanes (zero beyond the row's start): data-parallel primitives on the two
// halves of the double, no LDS round trip
template <int CTRL>
__device__ inline double g2_dpp_add(double v) {
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xf, 0xf, true);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xf, 0xf, true);
    return v + __hiloint2double(hi, lo);
}
// row sums of two values: afterwards lanes 15, 31, 47, 63 hold the sums of their 16 lanes
__device__ inline void g2_row_sum2(double& a, double& b) {
    a = g2_dpp_add<0x111>(a); b = g2_dpp_add<0x111>(b);   // row_shr:1
    a = g2_dpp_add<0x112>(a); b = g2_dpp_add<0x112>(b);   // row_shr:2
    a = g2_dpp_add<0x114>(a); b = g2_dpp_add<0x114>(b);   // row_shr:4
    a = g2_dpp_add<0x118>(a); b = g2_dpp_add<0x118>(b);   // row_shr:8
}

#ifndef TPHIP_GRAD2_MIN_WAVES
#define TPHIP_GRAD2_MIN_WAVES 3   // waves per SIMD the register allocation must allow (168 VGPRs)
#endif
template <int D>
__global__ __launch_bounds__(kGrad2Block, TPHIP_GRAD2_MIN_WAVES) void locus_grad2_kernel(const Grad2Params* __restrict__ Gp) {
    const Grad2Params& G = *Gp;
    extern __shared__ double lds[];
    const int nn = G.nnodes;
    double* TP = lds;                                              // [ntaxa][kValueTipRow] rows of P^T + a row of ones
    double* tipY = TP + (size_t)G.ntaxa * kValueTipRow;            // [16 masks][4] U^-1 (0/1 vector of the mask)
    double* astack = tipY + 64;                                    // [rdepth][4][kGrad2Block] parked adjoints
    double* gacc = astack + (size_t)G.rdepth * 4 * kGrad2Block;    // [waves][nn] sum over columns of w L'/L per branch
    double* hacc = gacc + (size_t)kGrad2Waves * nn;                // [waves][nn] sum of w (L''/L - (L'/L)^2)
    uint32_t* msk = (uint32_t*)(hacc + (size_t)kGrad2Waves * nn);  // [nwords][kGrad2Block] packed state codes of the thread's column
    __shared__ double eig[36];
    __shared__ double red[kGrad2Waves * 12];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const g2_gptr tape = (g2_gptr)(uintptr_t)G.tape + (size_t)blockIdx.x * (size_t)(G.ntape > 0 ? G.ntape : 1) * 4 * kGrad2Block + tid;
    const g2_gcu32 packed = (g2_gcu32)(uintptr_t)G.packed;
    double* gw = gacc + (size_t)wave * nn;
    double* hw = hacc + (size_t)wave * nn;
    const int64_t nitems = G.ncand * G.nsplit;
    for (int64_t item = blockIdx.x; item < nitems; item += gridDim.x) {
        __syncthreads();
        const int64_t cand = item / G.nsplit;
        const int locus = G.cand_locus[cand];
        int64_t lo, hi;
        split_range(G.locus_offsets[locus], G.locus_offsets[locus + 1], G.nsplit, (int)(item % G.nsplit), kGrad2Block, &lo, &hi);
        if (lo >= hi) {   // empty slice (uniform for the block): zero partials
            if (tid == 0) { G.out_lnl[item] = 0.0; G.out_sum_dlogt[item] = 0.0; }
            if (tid < 6) G.out_dexch[item * 6 + tid] = 0.0;
            if (G.out_dlogt) for (int b = tid; b < nn; b += kGrad2Block) G.out_dlogt[item * nn + b] = 0.0;
            if (G.out_d2logt) for (int b = tid; b < nn; b += kGrad2Block) G.out_d2logt[item * nn + b] = 0.0;
            continue;
        }
        if (tid < 36) eig[tid] = G.eig[(size_t)cand * 36 + tid];
        const double* pm = G.pmat + (size_t)cand * nn * 16;
        for (int t = tid; t < G.ntaxa; t += kGrad2Block) {   // the candidate's tip matrices -> LDS, by taxon
            const int node = G.tip_node[t];
            if (node >= 0) {
                const double2* src = (const double2*)(pm + (size_t)node * 16);
                double2* dst = (double2*)(TP + (size_t)t * kValueTipRow);
#pragma unroll
                for (int q = 0; q < 8; ++q) dst[q] = src[q];
                dst[8] = make_double2(1.0, 1.0); dst[9] = make_double2(1.0, 1.0);
            }
        }
        for (int b = tid; b < kGrad2Waves * nn; b += kGrad2Block) { gacc[b] = 0.0; hacc[b] = 0.0; }
        __syncthreads();
        if (tid < 64) {  // Y[mask] = U^-1 (0/1 vector of the state mask)
            const int m = tid >> 2, k = tid & 3;
            double s = 0;
#pragma unroll
            for (int j = 0; j < 4; ++j) s += ((m >> j) & 1) ? eig[20 + k * 4 + j] : 0.0;
            tipY[tid] = s;
        }
        __syncthreads();
        const double* pig = G.models[locus].pi;
        double U[16], lam[4], pi[4];
#pragma unroll
        for (int i = 0; i < 16; ++i) U[i] = lik_uniform(eig[4 + i]);
#pragma unroll
        for (int k = 0; k < 4; ++k) { lam[k] = lik_uniform(eig[k]); pi[k] = lik_uniform(pig[k]); }
        const value_cptr pmc = (value_cptr)(uintptr_t)pm;
        const g2_ops_cptr fops = (g2_ops_cptr)(uintptr_t)G.fops, rops = (g2_ops_cptr)(uintptr_t)G.rops;
        const value_cptr efc = (value_cptr)(uintptr_t)(G.ef + (size_t)cand * nn * kGrad2EF);
        double total = 0.0, inv_w = 0.0;
        double Wd[4] = {0, 0, 0, 0};           // sum_b t_b e_k x_k y_k
        double Ws[6] = {0, 0, 0, 0, 0, 0};     // sum_b F_kl (x_k y_l + x_l y_k), pairs 01 02 03 12 13 23

        // r = U^T z
        auto mulUt = [&](const double* z, double* r) {
#pragma unroll
            for (int k = 0; k < 4; ++k) r[k] = fma(U[12 + k], z[3], fma(U[8 + k], z[2], fma(U[4 + k], z[1], U[k] * z[0])));
        };
        // one branch: x = U^T (adjoint of its message), y = U^-1 (partial below it); W sums, and the per-branch sums of `node`
        auto contribute = [&](int node, const double* x, const double* y, double* e_out) {
            value_cptr ef = efc + (size_t)node * kGrad2EF;
            const double e0 = ef[0], e1 = ef[1], e2 = ef[2], e3 = ef[3], t = ef[10];
            const double f01 = ef[4], f02 = ef[5], f03 = ef[6], f12 = ef[7], f13 = ef[8], f23 = ef[9];
            if (e_out) { e_out[0] = e0; e_out[1] = e1; e_out[2] = e2; e_out[3] = e3; }
            const double d0 = x[0] * y[0] * e0, d1 = x[1] * y[1] * e1, d2 = x[2] * y[2] * e2, d3 = x[3] * y[3] * e3;
            Wd[0] = fma(t, d0, Wd[0]); Wd[1] = fma(t, d1, Wd[1]); Wd[2] = fma(t, d2, Wd[2]); Wd[3] = fma(t, d3, Wd[3]);
            Ws[0] = fma(f01, fma(x[0], y[1], x[1] * y[0]), Ws[0]);
            Ws[1] = fma(f02, fma(x[0], y[2], x[2] * y[0]), Ws[1]);
            Ws[2] = fma(f03, fma(x[0], y[3], x[3] * y[0]), Ws[2]);
            Ws[3] = fma(f12, fma(x[1], y[2], x[2] * y[1]), Ws[3]);
            Ws[4] = fma(f13, fma(x[1], y[3], x[3] * y[1]), Ws[4]);
            Ws[5] = fma(f23, fma(x[2], y[3], x[3] * y[2]), Ws[5]);
            const double p0 = lam[0] * d0, p1 = lam[1] * d1, p2 = lam[2] * d2, p3 = lam[3] * d3;
            double c = (p0 + p1) + (p2 + p3);                                                      // w L'/L of this column
            double h = fma(lam[3], p3, fma(lam[2], p2, fma(lam[1], p1, lam[0] * p0))) - c * c * inv_w;   // w (L''/L - (L'/L)^2)
            g2_row_sum2(c, h);
            if ((lane & 15) == 15) { atomicAdd(gw + node, c); atomicAdd(hw + node, h); }   // four lanes per wave, LDS
        };

        for (int64_t base = lo; base < hi; base += kGrad2Block) {
            const int64_t colx = base + tid;
            const bool active = colx < hi;
            const int64_t col = active ? colx : lo;
#pragma unroll 4
            for (int w = 0; w < G.nwords; ++w) msk[(size_t)w * kGrad2Block + tid] = packed[(size_t)w * G.ncols_total + col];
            // ---------------- forward (locus_value_kernel's sweep, one column per thread) ----------------
            double acc[4] = {1.0, 1.0, 1.0, 1.0};
            double stk[D][4];
#pragma unroll
            for (int d = 0; d < D; ++d)
#pragma unroll
                for (int i = 0; i < 4; ++i) stk[d][i] = 1.0;
            int scale = 0, sp = 0;
            unsigned word = 0;
            auto tip_msg = [&](int tab, unsigned cd, double* msg) {
                const char* tp = (const char*)TP + tab;
                if (!__any(cd > 4u)) {       // rows 0..3 = the resolved states, row 4 = ones (gap / N)
                    const double2* row = (const double2*)(tp + cd * 32);
                    const double2 r0 = row[0], r1 = row[1];
                    msg[0] = r0.x; msg[1] = r0.y; msg[2] = r1.x; msg[3] = r1.y;
                } else {                     // some lane holds an ambiguity code: the rows of its bits
                    const double* tpd = (const double*)tp;
                    const unsigned mk = (unsigned)((kValueMaskOfCode >> (4 * cd)) & 15ull);
#pragma unroll
                    for (int i = 0; i < 4; ++i) msg[i] = 0.0;
#pragma unroll
                    for (int x = 0; x < 4; ++x) {
                        const double f = ((mk >> x) & 1u) ? 1.0 : 0.0;
#pragma unroll
                        for (int i = 0; i < 4; ++i) msg[i] = fma(f, tpd[x * 4 + i], msg[i]);
                    }
                    // the lanes beside it keep exactly what the fast path gives them (wave-independent results)
                    const double2* row = (const double2*)(tp + (cd <= 4u ? cd : 0u) * 32);
                    const double2 r0 = row[0], r1 = row[1];
                    if (cd <= 4u) { msg[0] = r0.x; msg[1] = r0.y; msg[2] = r1.x; msg[3] = r1.y; }
                }
            };
            auto tip_op = [&](int tab, int sh, bool fetch, int widx, auto set_tag) {
                constexpr bool kSet = decltype(set_tag)::value;
                if (fetch) word = msk[(size_t)widx * kGrad2Block + tid];
                double msg[4];
                tip_msg(tab, (word >> sh) & 15u, msg);
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[i] = kSet ? msg[i] : acc[i] * msg[i];
            };
            double Pb[16];
#pragma unroll
            for (int q = 0; q < 16; ++q) Pb[q] = 0.0;
            g2_i4 op = fops[0], nxt = fops[1];
            for (int ip = 0; (op.x & OP_CODE_MASK) != kValueOpEnd; ++ip) {
                const g2_i4 nxt2 = fops[ip + 2];
                const int code = op.x & OP_CODE_MASK;
                const bool branch_next = (nxt.x & OP_CODE_MASK) == OP_BRANCH;
                if (code != OP_BRANCH && branch_next) {
                    value_cptr pb = (value_cptr)((const char __attribute__((address_space(4)))*)pmc + nxt.y);
#pragma unroll
                    for (int q = 0; q < 16; ++q) Pb[q] = pb[q];
                }
                if (code == OP_BRANCH) {
                    const unsigned hm = max(max((unsigned)__double2hiint(acc[0]), (unsigned)__double2hiint(acc[1])),
                                            max((unsigned)__double2hiint(acc[2]), (unsigned)__double2hiint(acc[3])));
                    if (__any(hm < 0x2B2BFF2Fu)) {   // high word of kLikRescaleBelow = 1e-100
                        const double mx = fmax(fmax(acc[0], acc[1]), fmax(acc[2], acc[3]));
                        if (mx < kLikRescaleBelow && mx > 0) {
                            int e;
                            frexp(mx, &e);
#pragma unroll
                            for (int i = 0; i < 4; ++i) acc[i] = ldexp(acc[i], -e);
                            scale += e;
                        }
                    }
                    const bool pop = (op.x & OP_POP_AFTER) != 0;
                    if (pop) --sp;
                    double n[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        n[i] = fma(Pb[12 + i], acc[3], fma(Pb[8 + i], acc[2], fma(Pb[4 + i], acc[1], Pb[i] * acc[0])));
                    const g2_gptr slot = tape + (size_t)op.z * 4 * kGrad2Block;   // the branch's message: what the reverse sweep reads
#pragma unroll
                    for (int i = 0; i < 4; ++i) slot[i * kGrad2Block] = n[i];
                    if (pop) {
#pragma unroll
                        for (int d = 0; d < D; ++d)
                            if (sp == d) {
#pragma unroll
                                for (int i = 0; i < 4; ++i) acc[i] = n[i] * stk[d][i];
                            }
                    } else {
#pragma unroll
                        for (int i = 0; i < 4; ++i) acc[i] = n[i];
                    }
                    if (branch_next) {
                        value_cptr pb = (value_cptr)((const char __attribute__((address_space(4)))*)pmc + nxt.y);
#pragma unroll
                        for (int q = 0; q < 16; ++q) Pb[q] = pb[q];
                    }
                } else {
                    if (op.x & OP_PUSH_BEFORE) {
#pragma unroll
                        for (int d = 0; d < D; ++d)
                            if (sp == d) {
#pragma unroll
                                for (int i = 0; i < 4; ++i) stk[d][i] = acc[i];
                            }
                        ++sp;
                    }
                    const int sh_a = (op.x >> 12) & 31, sh_b = (op.x >> 20) & 31;
                    const bool fetch_a = (op.x >> 17) & 1, fetch_b = (op.x >> 25) & 1;
                    if (code == OP_TIP_MUL) tip_op(op.y, sh_a, fetch_a, op.w & 0xffff, std::false_type{});
                    else tip_op(op.y, sh_a, fetch_a, op.w & 0xffff, std::true_type{});
                    if (code == OP_CHERRY) tip_op(op.z, sh_b, fetch_b, (op.w >> 16) & 0xffff, std::false_type{});
                }
                op = nxt;
                nxt = nxt2;
            }
            const double Lc = fma(pi[3], acc[3], fma(pi[2], acc[2], fma(pi[1], acc[1], pi[0] * acc[0])));
            const double cw = active ? (G.col_weight ? G.col_weight[col] : 1.0) : 0.0;
            total = fma(cw, log(Lc) + (double)scale * 0.6931471805599453, total);
            // ---------------- reverse: pre-order over the internal nodes ----------------
            const double seed = cw / Lc;   // padding lanes contribute exact zeros
            inv_w = cw > 0.0 ? 1.0 / cw : 0.0;
            double u[4] = {0.0, 0.0, 0.0, 0.0};
            int rsp = 0;
            g2_i4 ra = rops[0], rb = rops[1];
            // the messages of a record's internal children are requested one record ahead: the tape round trip (L2 for the
            // last few slots written, the memory-side cache or HBM for the rest) then runs under the previous record's arithmetic
            double nA[4] = {0.0, 0.0, 0.0, 0.0}, nB[4] = {0.0, 0.0, 0.0, 0.0};
            auto request = [&](const g2_i4& r4) {
                if (!(r4.x & G2_A_TIP)) {
                    const g2_gptr slot = tape + (size_t)r4.z * 4 * kGrad2Block;
#pragma unroll
                    for (int i = 0; i < 4; ++i) nA[i] = slot[i * kGrad2Block];
                }
                if (!(r4.x & G2_B_TIP)) {
                    const g2_gptr slot = tape + (size_t)r4.w * 4 * kGrad2Block;
#pragma unroll
                    for (int i = 0; i < 4; ++i) nB[i] = slot[i * kGrad2Block];
                }
            };
            request(ra);
            for (int ir = 0; ir < G.nrops; ++ir) {
                const g2_i4 a4 = ra, b4 = rb;
                double pA[4], pB[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) { pA[i] = nA[i]; pB[i] = nB[i]; }
                if (ir + 1 < G.nrops) {
                    ra = rops[2 * ir + 2]; rb = rops[2 * ir + 3];
                    request(ra);
                }
                const int flags = a4.x;
                if (flags & G2_POP_U) {
                    --rsp;
                    const double* as_ = astack + ((size_t)rsp * 4) * kGrad2Block + tid;
#pragma unroll
                    for (int i = 0; i < 4; ++i) u[i] = as_[i * kGrad2Block];
                }
                double mA[4], mB[4];
                unsigned cdA = 0, cdB = 0;
                if (flags & G2_A_TIP) {
                    cdA = (msk[(size_t)(b4.x & 0xffff) * kGrad2Block + tid] >> (b4.x >> 16)) & 15u;
                    tip_msg(a4.z, cdA, mA);
                } else {
#pragma unroll
                    for (int i = 0; i < 4; ++i) mA[i] = pA[i];   // requested while the previous record was worked on
                }
                if (flags & G2_B_TIP) {
                    cdB = (msk[(size_t)(b4.y & 0xffff) * kGrad2Block + tid] >> (b4.y >> 16)) & 15u;
                    tip_msg(a4.w, cdB, mB);
                } else {
#pragma unroll
                    for (int i = 0; i < 4; ++i) mB[i] = pB[i];
                }
                double a[4];
                if (flags & G2_IS_ROOT) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) a[i] = pi[i] * seed;
                } else {
                    double l[4], w[4], x[4], y[4], e[4], z[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) l[i] = mA[i] * mB[i];
                    // the forward sweep's rescale of this partial, redone (its factor belongs to the message: m = P (sc l))
                    double sc = 1.0;
                    const unsigned hm = max(max((unsigned)__double2hiint(l[0]), (unsigned)__double2hiint(l[1])),
                                            max((unsigned)__double2hiint(l[2]), (unsigned)__double2hiint(l[3])));
                    if (__any(hm < 0x2B2BFF2Fu)) {
                        const double mx = fmax(fmax(l[0], l[1]), fmax(l[2], l[3]));
                        if (mx < kLikRescaleBelow && mx > 0) {
                            int ex;
                            frexp(mx, &ex);
#pragma unroll
                            for (int i = 0; i < 4; ++i) l[i] = ldexp(l[i], -ex);
                            sc = ldexp(1.0, -ex);
                        }
                    }
#pragma unroll
                    for (int i = 0; i < 4; ++i) w[i] = pi[i] * l[i];
                    mulUt(w, y);
                    mulUt(u, x);
                    contribute(a4.y, x, y, e);
#pragma unroll
                    for (int k = 0; k < 4; ++k) z[k] = e[k] * x[k];
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        a[i] = sc * pi[i] * fma(U[i * 4 + 3], z[3], fma(U[i * 4 + 2], z[2], fma(U[i * 4 + 1], z[1], U[i * 4] * z[0])));
                }
                double uA[4], uB[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) { uA[i] = a[i] * mB[i]; uB[i] = a[i] * mA[i]; }
                if (flags & G2_A_TIP) {
                    double x[4];
                    mulUt(uA, x);
                    const unsigned mk = (unsigned)((kValueMaskOfCode >> (4 * cdA)) & 15ull);
                    const double2 y01 = *(const double2*)(tipY + mk * 4), y23 = *(const double2*)(tipY + mk * 4 + 2);
                    const double y[4] = {y01.x, y01.y, y23.x, y23.y};
                    contribute(b4.z & 0xffff, x, y, nullptr);
                }
                if (flags & G2_B_TIP) {
                    double x[4];
                    mulUt(uB, x);
                    const unsigned mk = (unsigned)((kValueMaskOfCode >> (4 * cdB)) & 15ull);
                    const double2 y01 = *(const double2*)(tipY + mk * 4), y23 = *(const double2*)(tipY + mk * 4 + 2);
                    const double y[4] = {y01.x, y01.y, y23.x, y23.y};
                    contribute((b4.z >> 16) & 0xffff, x, y, nullptr);
                }
                if (flags & G2_PUSH_B) {
                    double* as_ = astack + ((size_t)rsp * 4) * kGrad2Block + tid;
#pragma unroll
                    for (int i = 0; i < 4; ++i) as_[i * kGrad2Block] = uB[i];
                    ++rsp;
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) u[i] = uA[i];   // carried when A is internal (its record is next)
            }
        }
        // ---------------- reductions (fixed order) ----------------
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            total += __shfl_xor(total, o);
#pragma unroll
            for (int i = 0; i < 4; ++i) Wd[i] += __shfl_xor(Wd[i], o);
#pragma unroll
            for (int i = 0; i < 6; ++i) Ws[i] += __shfl_xor(Ws[i], o);
        }
        if (lane == 0) {
            red[wave * 12] = total;
            for (int i = 0; i < 4; ++i) red[wave * 12 + 1 + i] = Wd[i];
            for (int i = 0; i < 6; ++i) red[wave * 12 + 5 + i] = Ws[i];
        }
        __syncthreads();
        if (tid == 0) {
            double lnl = 0, wd[4] = {0, 0, 0, 0}, ws[6] = {0, 0, 0, 0, 0, 0};
            for (int w = 0; w < kGrad2Waves; ++w) {
                lnl += red[w * 12];
                for (int i = 0; i < 4; ++i) wd[i] += red[w * 12 + 1 + i];
                for (int i = 0; i < 6; ++i) ws[i] += red[w * 12 + 5 + i];
            }
            G.out_lnl[item] = lnl;
            // dQ/dr_ij = pi_j (E_ij - E_ii) + pi_i (E_ji - E_jj)  ->  G_kl = (U_jl - U_il) (pi_j Ui_ki - pi_i Ui_kj), symmetric
            const int pi_[6] = {0, 0, 0, 1, 1, 2}, pj_[6] = {1, 2, 3, 2, 3, 3};
            const int pk_[6] = {0, 0, 0, 1, 1, 2}, pl_[6] = {1, 2, 3, 2, 3, 3};
            for (int q = 0; q < 6; ++q) {
                const int i = pi_[q], j = pj_[q];
                double d = 0;
                for (int k = 0; k < 4; ++k)
                    d += wd[k] * (eig[4 + j * 4 + k] - eig[4 + i * 4 + k]) * (pig[j] * eig[20 + k * 4 + i] - pig[i] * eig[20 + k * 4 + j]);
                for (int s = 0; s < 6; ++s) {
                    const int k = pk_[s], l = pl_[s];
                    d += ws[s] * (eig[4 + j * 4 + l] - eig[4 + i * 4 + l]) * (pig[j] * eig[20 + k * 4 + i] - pig[i] * eig[20 + k * 4 + j]);
                }
                G.out_dexch[item * 6 + q] = d;
            }
        }
        // d lnL / d log t_b = t_b * sum over waves; and its total
        double part = 0.0;
        for (int b = tid; b < nn; b += kGrad2Block) {
            double g = 0, h = 0;
            for (int w = 0; w < kGrad2Waves; ++w) { g += gacc[(size_t)w * nn + b]; h += hacc[(size_t)w * nn + b]; }
            const double t = G.ef[((size_t)cand * nn + b) * kGrad2EF + 10];
            if (G.out_d2logt) G.out_d2logt[item * nn + b] = t * (t * h + g);   // d2/d(log t)^2 = t^2 d2/dt^2 + t d/dt
            g *= t;
            if (G.out_dlogt) G.out_dlogt[item * nn + b] = g;
            part += g;
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) part += __shfl_xor(part, o);
        __syncthreads();   // red[] was read by thread 0 above
        if (lane == 0) red[wave] = part;
        __syncthreads();
        if (tid == 0) {
            double s = 0;
            for (int w = 0; w < kGrad2Waves; ++w) s += red[w];
            G.out_sum_dlogt[item] = s;
        }
    }
}

}  // namespace tphip
