// locus_value_kernel.hpp -- whole-locus log-likelihood, value only, from per-branch transition matrices.
//
// The objective behind every `Optimize(lf_MLES, lf)` of HyPhy's stage 1 (tapir/data/models_and_rates.bf:487-520,
// 647-655): sum over a locus' columns of log L(column | exchangeabilities, branch lengths), all sites at rate 1.
// Stage 1 asks for it millions of times (line searches of the general model, difference stencils and the 202
// rate-class models of every locus), so this is the kernel stage 1's wall time is made of.
//
// Without a per-site rate every column of a candidate sees the SAME transition matrix on a branch.  Three kernels:
//
//   lik_eigen_kernel   thread = candidate          Q = R o pi diagonalised (4x4 cyclic Jacobi) -> [36] doubles.
//                                                  (Inside locus_loglik_kernel this was thread 0 of every workgroup: ~6000
//                                                  single-lane instructions, more than the pruning of a 300-pattern locus.)
//   lik_pmat_kernel    thread = (candidate, node)  P_b = U exp(Lambda t_b) U^-1, stored transposed ([child state][parent
//                                                  state]) so that the message of a resolved tip is one 32-byte row.
//   locus_value_kernel workgroup = (candidate, column slice), thread = C columns
//        tip      acc *= row x of P_b from an LDS copy of the candidate's tip matrices (2 ds_read_b128 + 4 multiplies).  Tip
//                 states travel as 4-bit CODES (locus_value_params.hpp): 0..3 = the row of a resolved state, 4 = a row of ones
//                 (gap / N: the row sums of P), 5..14 = the other IUPAC sets, which add the rows of their bits on a slow path
//                 taken only by waves that hold one.  The thread's codes are packed 8 per word into thread-private LDS rows
//                 at the start of a column group, 16 tips' loads in flight together, instead of one memory round trip per tip op
//        branch   acc  = P_b acc with P_b in SCALAR registers (one 128-byte scalar load per op and wave, requested while the
//                 op before it runs): 16 FP64 instructions per column instead of the 40 of the eigenbasis form
//                 U (e_b o U^T (pi o acc)); the rescale test is 4 integer instructions on the exponent fields and a ballot
//        parked siblings in REGISTERS (the stack depth D is a template parameter: 3 slots on a 64-taxon tree), so LDS holds
//        only the tip matrices and code words and occupancy is set by registers (111 VGPRs for C = 2, D = 3: 4 waves per SIMD)
//        op stream: TIP_SET + TIP_MUL pairs fused into CHERRY, PUSH / POP_MUL riding as flags on their neighbours
//        (105 interpreter iterations instead of 189 on 64 taxa), every field precomputed in the 16-byte record, records
//        fetched two ahead, decoded once for the thread's C columns.
//
// Bound: FP64 VALU, ~24 ntaxa useful instructions per column and candidate (the eigenbasis kernel: ~83 ntaxa).  The
// matrices cost nnodes * 128 B per candidate of workspace, written once and read once per wave and column group:
// candidates are processed in chunks that keep the workspace under kValueWorkspaceBytes.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include <type_traits>
#include "gtr_model.hpp"
#include "tree_program.hpp"
#include "locus_lik_common.hpp"
#include "locus_value_params.hpp"

namespace tphip {

// eig[cand] = lam[4], U[16], U^-1[16] (lik_eigen's layout)
__global__ __launch_bounds__(64) void lik_eigen_kernel(const LocusModel* models, const int32_t* cand_locus, const double* cand_exch,
                                                       int64_t ncand, double* eig_out) {
    const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= ncand) return;
    double eig[36];
    lik_eigen(models[cand_locus[c]].pi, cand_exch + (size_t)c * 6, eig);
#pragma unroll
    for (int i = 0; i < 36; ++i) eig_out[(size_t)c * 36 + i] = eig[i];
}

// pmat[cand][node][x][i] = P_b[i][x] = sum_k U[i][k] exp(lam_k t) U^-1[k][x], floored at kLikTiny (for a near-zero
// branch the eigen-sum of an off-diagonal entry cancels to rounding noise around 0, which must not turn a partial negative)
__global__ __launch_bounds__(128) void lik_pmat_kernel(const double* eig, const double* blen_vecs, const int32_t* cand_vec,
                                                       const double* cand_scale, const int32_t* cand_pidx, const double* cand_pfac,
                                                       int64_t ncand, int32_t nnodes, double* pmat) {
    // A thread's matrix is 128 contiguous bytes and the block's 128 matrices one contiguous 16 KB span: written from the
    // registers, every store instruction touched 64 different 128-byte lines (1.2 TB/s on 27 GB per pass of 2000 loci);
    // through an LDS tile (rows padded to 17 doubles) every store instruction writes 512 contiguous bytes.
    __shared__ double tile[128 * 17];
    const int64_t total = ncand * nnodes;
    const int64_t base = (int64_t)blockIdx.x * blockDim.x;
    const int64_t idx = base + threadIdx.x;
    if (idx < total) {
        const int64_t c = idx / nnodes;
        const int b = (int)(idx - c * nnodes);
        const double* E = eig + (size_t)c * 36;
        const double t = blen_vecs[(size_t)cand_vec[c] * nnodes + b] * cand_scale[c] * (b == cand_pidx[c] ? cand_pfac[c] : 1.0);
        double e[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) e[k] = exp(E[k] * t);
        double* row = tile + threadIdx.x * 17;
#pragma unroll
        for (int x = 0; x < 4; ++x)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                double s = 0.0;
#pragma unroll
                for (int k = 0; k < 4; ++k) s = fma(E[4 + i * 4 + k] * e[k], E[20 + k * 4 + x], s);
                row[x * 4 + i] = fmax(s, kLikTiny);
            }
    }
    __syncthreads();
    const int64_t left = total - base;
    const int nvals = (int)(left < (int64_t)blockDim.x ? left : (int64_t)blockDim.x) * 16;
    double* out = pmat + (size_t)base * 16;
    for (int q = threadIdx.x; q < nvals; q += blockDim.x) out[q] = tile[(q >> 4) * 17 + (q & 15)];
}

// packed[w][col] = the state codes of tips 8w .. 8w+7 (program order) of column col, 4 bits each
__global__ __launch_bounds__(256) void value_pack_codes_kernel(const uint8_t* states, int64_t ncols_total, const int32_t* tip_taxon,
                                                               int32_t nwords, uint32_t* packed) {
    const int64_t col = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (col >= ncols_total) return;
    for (int w = 0; w < nwords; ++w) {
        unsigned word = 0;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const unsigned m = states[(int64_t)tip_taxon[w * 8 + q] * ncols_total + col] & 15u;
            word |= (unsigned)((kValueCodeOfMask >> (4 * m)) & 15ull) << (4 * q);
        }
        packed[(size_t)w * ncols_total + col] = word;
    }
}

typedef const double __attribute__((address_space(4)))* value_cptr;   // constant address space: wave-uniform reads become scalar loads

#ifndef TPHIP_VALUE_MIN_WAVES
#define TPHIP_VALUE_MIN_WAVES 4   // waves per SIMD the register allocation must allow: 128 VGPRs, 14 of the two-column
                                 // variant's registers in scratch -- measured 10 % faster than 143 VGPRs at 3 waves (stage 1, 2000 x 1000 x 64)
#endif
template <int C, int D>
__global__ __launch_bounds__(kLikBlock, TPHIP_VALUE_MIN_WAVES) void locus_value_kernel(ValueParams P) {
    extern __shared__ double lds[];
    double* TP = lds;                                           // [ntaxa][4 child states + a row of ones][4 parent states]
    uint32_t* msk = (uint32_t*)(TP + (size_t)P.ntaxa * kValueTipRow);   // [nwords][C][kLikBlock] 8 tip codes per word
    __shared__ double red[kLikBlock / 64];
    const int64_t cand = blockIdx.x / P.nsplit;
    const int locus = P.cand_locus[cand];
    int64_t lo, hi;
    split_range(P.locus_offsets[locus], P.locus_offsets[locus + 1], P.nsplit, (int)(blockIdx.x % P.nsplit), kLikBlock * C, &lo, &hi);
    if (lo >= hi) {   // empty slice (uniform for the block)
        if (threadIdx.x == 0) P.out[blockIdx.x] = 0.0;
        return;
    }
    const double* pm = P.pmat + (size_t)cand * P.nnodes * 16;
    for (int t = threadIdx.x; t < P.ntaxa; t += kLikBlock) {   // the candidate's tip matrices -> LDS, by taxon
        const int node = P.tip_node[t];
        if (node >= 0) {
            const double2* src = (const double2*)(pm + (size_t)node * 16);
            double2* dst = (double2*)(TP + (size_t)t * kValueTipRow);
#pragma unroll
            for (int q = 0; q < 8; ++q) dst[q] = src[q];
            dst[8] = make_double2(1.0, 1.0); dst[9] = make_double2(1.0, 1.0);
        }
    }
    __syncthreads();
    const double* pig = P.models[locus].pi;
    double pi[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) pi[k] = lik_uniform(pig[k]);
    const value_cptr pmc = (value_cptr)(uintptr_t)pm;
    double total = 0.0;
    for (int64_t base = lo; base < hi; base += (int64_t)kLikBlock * C) {
        int64_t col[C];
        bool active[C];
        double acc[C][4];
        double stk[D][C][4];
        int scale[C];
#pragma unroll
        for (int c = 0; c < C; ++c) {
            const int64_t x = base + (int64_t)c * kLikBlock + threadIdx.x;
            active[c] = x < hi;
            col[c] = active[c] ? x : lo;
            scale[c] = 0;
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[c][i] = 1.0;
        }
#pragma unroll
        for (int d = 0; d < D; ++d)
#pragma unroll
            for (int c = 0; c < C; ++c)
#pragma unroll
                for (int i = 0; i < 4; ++i) stk[d][c][i] = 1.0;
        // The thread's state codes, in tip order, 8 per word, into thread-private rows of LDS (no barrier): from the array
        // packed at upload time when there is one (one dword per 8 tips), else from the state bytes with the loads of two
        // words in flight together.
        if (P.packed) {
#pragma unroll 4
            for (int w = 0; w < P.nwords; ++w)
#pragma unroll
                for (int c = 0; c < C; ++c) msk[((size_t)w * C + c) * kLikBlock + threadIdx.x] = P.packed[(size_t)w * P.ncols_total + col[c]];
        } else {
            for (int w0 = 0; w0 < P.nwords; w0 += 2) {
                unsigned b[C][16];
#pragma unroll
                for (int q = 0; q < 16; ++q) {
                    const int j = w0 * 8 + q;
                    const int taxon = P.tip_taxon[j < P.nwords * 8 ? j : 0];
#pragma unroll
                    for (int c = 0; c < C; ++c) b[c][q] = P.states[(int64_t)taxon * P.ncols_total + col[c]];
                }
#pragma unroll
                for (int c = 0; c < C; ++c) {
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        unsigned word = 0;
#pragma unroll
                        for (int q = 0; q < 8; ++q)
                            word |= (unsigned)((kValueCodeOfMask >> (4 * (b[c][h * 8 + q] & 15u))) & 15ull) << (4 * q);
                        if (w0 + h < P.nwords) msk[((size_t)(w0 + h) * C + c) * kLikBlock + threadIdx.x] = word;
                    }
                }
            }
        }
        int sp = 0;
        unsigned word[C];
        // one tip: acc (*)= row(s) of the tip's transition matrix selected by its state code.  `tab` = byte offset of the
        // taxon's table, `sh` = bit position of the tip's code in its word, `fetch` = the tip opens a new word (index widx)
        auto tip_op = [&](int tab, int sh, bool fetch, int widx, auto set_tag) {
            constexpr bool kSet = decltype(set_tag)::value;   // TIP_SET: the subtree starts here (program start or right after a PUSH)
            if (fetch) {
#pragma unroll
                for (int c = 0; c < C; ++c) word[c] = msk[((size_t)widx * C + c) * kLikBlock + threadIdx.x];
            }
            const char* tp = (const char*)TP + tab;
            unsigned cd[C];
            bool amb = false;
#pragma unroll
            for (int c = 0; c < C; ++c) {
                cd[c] = (word[c] >> sh) & 15u;
                amb |= cd[c] > 4u;
            }
            double msg[C][4];
            if (!__any(amb)) {       // rows 0..3 = the resolved states, row 4 = ones (gap / N: the row sums of P)
#pragma unroll
                for (int c = 0; c < C; ++c) {
                    const double2* row = (const double2*)(tp + cd[c] * 32);
                    const double2 r0 = row[0], r1 = row[1];
                    msg[c][0] = r0.x; msg[c][1] = r0.y; msg[c][2] = r1.x; msg[c][3] = r1.y;
                }
            } else {                 // some lane holds an ambiguity code: the rows of its bits
                const double* tpd = (const double*)tp;
#pragma unroll
                for (int c = 0; c < C; ++c) {
                    const unsigned mk = (unsigned)((kValueMaskOfCode >> (4 * cd[c])) & 15ull);
#pragma unroll
                    for (int i = 0; i < 4; ++i) msg[c][i] = 0.0;
#pragma unroll
                    for (int x = 0; x < 4; ++x) {
                        const double f = ((mk >> x) & 1u) ? 1.0 : 0.0;
#pragma unroll
                        for (int i = 0; i < 4; ++i) msg[c][i] = fma(f, tpd[x * 4 + i], msg[c][i]);
                    }
                    // the lanes beside it keep EXACTLY what the fast path gives them (a gap is the row of ones, not the sum of
                    // four floored rows): a column's value must not depend on which columns share its wave
                    const double2* row = (const double2*)(tp + (cd[c] <= 4u ? cd[c] : 0u) * 32);
                    const double2 r0 = row[0], r1 = row[1];
                    if (cd[c] <= 4u) { msg[c][0] = r0.x; msg[c][1] = r0.y; msg[c][2] = r1.x; msg[c][3] = r1.y; }
                }
            }
#pragma unroll
            for (int c = 0; c < C; ++c)
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[c][i] = kSet ? msg[c][i] : acc[c][i] * msg[c][i];
        };
        // A BRANCH's matrix (32 scalar registers) is requested while the op BEFORE it runs: at the top of a tip op, whose
        // own LDS round trip then covers the scalar load's, or after the multiply-adds of a BRANCH that is followed by another.
        double Pb[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) Pb[q] = 0.0;
        // ops are fetched two ahead (the stream ends with two OP_END records): the scalar load issued at the top of an
        // iteration is first needed at the top of the next one
        int4 op = P.vops[0], nxt = P.vops[1];
        for (int ip = 0; (op.x & OP_CODE_MASK) != kValueOpEnd; ++ip) {
            const int4 nxt2 = P.vops[ip + 2];
            const int code = op.x & OP_CODE_MASK;
            const bool branch_next = (nxt.x & OP_CODE_MASK) == OP_BRANCH;
            if (code != OP_BRANCH && branch_next) {
                value_cptr pb = (value_cptr)((const char __attribute__((address_space(4)))*)pmc + nxt.y);
#pragma unroll
                for (int q = 0; q < 16; ++q) Pb[q] = pb[q];
            }
            if (code == OP_BRANCH) {
                // rescale test on the exponent fields (all entries are positive): 3 integer instructions and a ballot
                bool low = false;
#pragma unroll
                for (int c = 0; c < C; ++c) {
                    const unsigned hm = max(max((unsigned)__double2hiint(acc[c][0]), (unsigned)__double2hiint(acc[c][1])),
                                            max((unsigned)__double2hiint(acc[c][2]), (unsigned)__double2hiint(acc[c][3])));
                    low |= hm < 0x2B2BFF2Fu;   // high word of kLikRescaleBelow = 1e-100
                }
                if (__any(low)) {
#pragma unroll
                    for (int c = 0; c < C; ++c) {
                        const double mx = fmax(fmax(acc[c][0], acc[c][1]), fmax(acc[c][2], acc[c][3]));
                        if (mx < kLikRescaleBelow && mx > 0) {
                            int e;
                            frexp(mx, &e);
#pragma unroll
                            for (int i = 0; i < 4; ++i) acc[c][i] = ldexp(acc[c][i], -e);
                            scale[c] += e;
                        }
                    }
                }
                const bool pop = (op.x & OP_POP_AFTER) != 0;
                if (pop) --sp;
                double n[C][4];
#pragma unroll
                for (int c = 0; c < C; ++c)
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        n[c][i] = fma(Pb[12 + i], acc[c][3], fma(Pb[8 + i], acc[c][2], fma(Pb[4 + i], acc[c][1], Pb[i] * acc[c][0])));
                if (pop) {   // the product with the parked sibling writes the accumulator: no copy of n
#pragma unroll
                    for (int d = 0; d < D; ++d)
                        if (sp == d) {
#pragma unroll
                            for (int c = 0; c < C; ++c)
#pragma unroll
                                for (int i = 0; i < 4; ++i) acc[c][i] = n[c][i] * stk[d][c][i];
                        }
                } else {
#pragma unroll
                    for (int c = 0; c < C; ++c)
#pragma unroll
                        for (int i = 0; i < 4; ++i) acc[c][i] = n[c][i];
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
                            for (int c = 0; c < C; ++c)
#pragma unroll
                                for (int i = 0; i < 4; ++i) stk[d][c][i] = acc[c][i];
                        }
                    ++sp;
                }
                const int sh_a = (op.x >> 12) & 31, sh_b = (op.x >> 20) & 31;
                const bool fetch_a = (op.x >> 17) & 1, fetch_b = (op.x >> 25) & 1;
                if (code == OP_TIP_MUL) tip_op(op.y, sh_a, fetch_a, op.w & 0xffff, std::false_type{});
                else tip_op(op.y, sh_a, fetch_a, op.w & 0xffff, std::true_type{});   // TIP_SET / CHERRY: the first tip of a subtree
                if (code == OP_CHERRY) tip_op(op.z, sh_b, fetch_b, (op.w >> 16) & 0xffff, std::false_type{});
            }
            op = nxt;
            nxt = nxt2;
        }
#pragma unroll
        for (int c = 0; c < C; ++c) {
            const double L = fma(pi[3], acc[c][3], fma(pi[2], acc[c][2], fma(pi[1], acc[c][1], pi[0] * acc[c][0])));
            if (active[c]) {
                const double w = P.col_weight ? P.col_weight[col[c]] : 1.0;
                total = fma(w, log(L) + (double)scale[c] * 0.6931471805599453, total);
            }
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

}  // namespace tphip
