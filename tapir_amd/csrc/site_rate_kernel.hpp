// site_rate_kernel.hpp -- per-column substitution-rate ML (HyPhy stage 2) on gfx950.
//
// Reference behaviour replaced: tapir/data/models_and_rates.bf:1042-1070 -- for each alignment column
// `Optimize(site_res, siteLikelihood)` over the single scalar siteRate that multiplies every branch length
// of the fixed tree (bf:1003-1013), GTR model (bf:978-1001), start value 1 (bf:1050).
//
// Mapping to CDNA4
//   * one alignment column per lane, 64 columns (one wavefront) per workgroup, all of one locus, so the
//     locus' eigen-system and the tree program are wave-uniform: they live in SGPRs / the scalar cache and
//     enter the FP64 VALU instructions as scalar operands;
//   * the traversal is the uniform op stream of tree_program.hpp -- no lane divergence inside a likelihood
//     evaluation; lanes diverge only in how many Newton steps they need;
//   * the running partial (value, d/du, d2/du2 of the 4 conditional likelihoods; u = log siteRate) is 12
//     doubles in VGPRs; parked siblings go to an LDS stack laid out [slot][component][lane] so every
//     ds_read/write_b64 is a conflict-free 512-B row;
//   * tips: the per-mask vectors U^-1 * tip(mask) are a 16x4 table in LDS built once per workgroup;
//   * constant / flat columns never reach this kernel: classify_kernel answers them in closed form and
//     compact_kernel packs the remaining columns so that lanes are not idle beside them;
//   * MFMA is deliberately not used: the work is 4x4 mat-vecs with a different matrix (rate) per lane.
//     The kernel is FP64-VALU/transcendental bound; its HBM traffic is ntaxa+24 bytes per column.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include "tphip.h"
#include "gtr_model.hpp"
#include "tree_program.hpp"

namespace tphip {

constexpr double kUMin = -23.025850929940457;  // log(1e-10)
constexpr double kUMax = 9.210340371976184;    // log(1e4)
constexpr double kStepMax = 2.0;
constexpr double kStepTol = 1e-9;
constexpr int kMaxIt = 100;
constexpr double kFlatEps = 1e-10;  // |g| and |h| below this: log L flat to fp64 resolution -> saturated
constexpr int kSiteBlock = 64;  // one wavefront per workgroup

struct SiteParams {
    const uint8_t* states;       // [ntaxa][ncols_total]
    int64_t ncols_total;
    const LocusModel* models;    // [nloci]
    const TreeOp* ops;
    int32_t nops;
    int32_t stack_depth;
    double chrono_length;
    const int64_t* locus_offsets;  // [nloci+1]
    const int32_t* chunk_locus;    // [nchunks]  64-column chunks, never straddling loci
    const int32_t* chunk_index;    // [nchunks]  index of the chunk inside its locus
    const int32_t* work_cols;      // [ncols_total] compacted column ids, per locus at locus_offsets[l]
    const int32_t* work_count;     // [nloci]
    double* rate;
    double* subst;
    double* lnl;
    uint8_t* flag;
    unsigned long long* eval_counter;
};

struct Partial {  // value and first/second derivative (wrt u) of the 4 conditional likelihoods
    double v[4], d1[4], d2[4];
};

// exp(x) for x <= 0 (branch exponents lam*t*s are never positive).  Cody-Waite reduction + degree-11
// Taylor/minimax-equivalent polynomial on |r| <= ln2/2; 2^n applied through the exponent field.
// Max observed error vs libm < 1 ulp-ish on [-745, 0]; results below 2^-1022 flush to 0.
__device__ __forceinline__ double exp_nonpos(double x) {
    const double LOG2E = 1.4426950408889634074;
    const double LN2_HI = 6.93147180369123816490e-01;
    const double LN2_LO = 1.90821492927058770002e-10;
    double n = rint(x * LOG2E);
    double r = fma(-n, LN2_HI, x);
    r = fma(-n, LN2_LO, r);
    // exp(r) = 1 + r + r^2/2! + ... + r^13/13!   (|r| <= 0.3466: truncation < 2e-18)
    double p = 1.6059043836821613e-10;          // 1/13!
    p = fma(p, r, 2.08767569878681e-09);        // 1/12!
    p = fma(p, r, 2.505210838544172e-08);       // 1/11!
    p = fma(p, r, 2.755731922398589e-07);       // 1/10!
    p = fma(p, r, 2.7557319223985893e-06);      // 1/9!
    p = fma(p, r, 2.48015873015873e-05);        // 1/8!
    p = fma(p, r, 1.984126984126984e-04);       // 1/7!
    p = fma(p, r, 1.388888888888889e-03);       // 1/6!
    p = fma(p, r, 8.333333333333333e-03);       // 1/5!
    p = fma(p, r, 4.1666666666666664e-02);      // 1/4!
    p = fma(p, r, 1.6666666666666666e-01);      // 1/3!
    p = fma(p, r, 0.5);
    p = fma(p, r, 1.0);
    p = fma(p, r, 1.0);
    int ni = (int)n;
    if (ni < -1021) return 0.0;  // below the normal range: contributes nothing to any likelihood here
    // multiply by 2^ni through the exponent bits (p in [0.70, 1.42], result stays normal)
    long long bits = __double_as_longlong(p) + ((long long)ni << 52);
    return __longlong_as_double(bits);
}

// acc *= m (product rule for value / first / second derivative)
__device__ __forceinline__ void partial_mul(Partial& a, const Partial& m) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        double a0 = a.v[i], a1 = a.d1[i], a2 = a.d2[i];
        double t = a1 * m.d1[i];
        a.d2[i] = fma(a2, m.v[i], fma(a0, m.d2[i], t + t));
        a.d1[i] = fma(a1, m.v[i], a0 * m.d1[i]);
        a.v[i] = a0 * m.v[i];
    }
}

// message of a tip through its branch: P(t s) * tip, with derivatives wrt u = log s.
// w[k] = (U^-1 tip)_k from the LDS mask table; x_k = lam_k t s; e_k = exp(x_k).
__device__ __forceinline__ void tip_message(const LocusModel* __restrict__ M, const double* w, double ts, Partial& m) {
    double a[3], b[3], c[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        double x = M->lam[k + 1] * ts;
        double e = exp_nonpos(x);
        a[k] = e * w[k + 1];
        b[k] = x * a[k];
        c[k] = fma(x, b[k], b[k]);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const double* Ur = M->U + i * 4;
        m.v[i] = fma(Ur[3], a[2], fma(Ur[2], a[1], fma(Ur[1], a[0], w[0])));
        m.d1[i] = fma(Ur[3], b[2], fma(Ur[2], b[1], Ur[1] * b[0]));
        m.d2[i] = fma(Ur[3], c[2], fma(Ur[2], c[1], Ur[1] * c[0]));
    }
}

// acc <- P(t s) * acc with derivatives (internal branch)
__device__ __forceinline__ void branch_apply(const LocusModel* __restrict__ M, double ts, Partial& p) {
    double w0[4], w1[4], w2[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const double* Ir = M->Ui + k * 4;
        w0[k] = fma(Ir[3], p.v[3], fma(Ir[2], p.v[2], fma(Ir[1], p.v[1], Ir[0] * p.v[0])));
        w1[k] = fma(Ir[3], p.d1[3], fma(Ir[2], p.d1[2], fma(Ir[1], p.d1[1], Ir[0] * p.d1[0])));
        w2[k] = fma(Ir[3], p.d2[3], fma(Ir[2], p.d2[2], fma(Ir[1], p.d2[1], Ir[0] * p.d2[0])));
    }
    double a[3], b[3], c[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        double x = M->lam[k + 1] * ts;
        double e = exp_nonpos(x);
        a[k] = e * w0[k + 1];
        double ew1 = e * w1[k + 1], ew2 = e * w2[k + 1];
        b[k] = fma(x, a[k], ew1);
        c[k] = fma(x, b[k] + a[k] + ew1, ew2);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const double* Ur = M->U + i * 4;
        p.v[i] = fma(Ur[3], a[2], fma(Ur[2], a[1], fma(Ur[1], a[0], w0[0])));
        p.d1[i] = fma(Ur[3], b[2], fma(Ur[2], b[1], fma(Ur[1], b[0], w1[0])));
        p.d2[i] = fma(Ur[3], c[2], fma(Ur[2], c[1], fma(Ur[1], c[0], w2[0])));
    }
}

// Rescale when the partial gets small (deep trees / hundreds of taxa); exponent goes to `scale`.
__device__ __forceinline__ void partial_rescale(Partial& p, int& scale) {
    double mx = fmax(fmax(p.v[0], p.v[1]), fmax(p.v[2], p.v[3]));
    int e = (int)((__double_as_longlong(mx) >> 52) & 0x7ff) - 1023;
    if (mx > 0.0 && e < -256) {
        double f = __longlong_as_double((long long)(1023 - e) << 52);  // 2^-e
#pragma unroll
        for (int i = 0; i < 4; ++i) { p.v[i] *= f; p.d1[i] *= f; p.d2[i] *= f; }
        scale += e;
    }
}

// One likelihood evaluation for this lane's column: f = log L, g = df/du, h = d2f/du2 at u (s = exp(u)).
__device__ __forceinline__ void evaluate_column(const SiteParams& P, const LocusModel* __restrict__ M,
                                                const double* __restrict__ wtab, double* __restrict__ stack,
                                                int64_t col, double s, double& f, double& g, double& h) {
    Partial acc;
    int scale = 0;
    int sp = 0;
    const int lane = threadIdx.x;
#pragma unroll
    for (int i = 0; i < 4; ++i) { acc.v[i] = 1.0; acc.d1[i] = 0.0; acc.d2[i] = 0.0; }
    for (int ip = 0; ip < P.nops; ++ip) {
        const TreeOp op = P.ops[ip];  // uniform address -> scalar load
        if (op.code <= OP_TIP_MUL) {
            unsigned mask = P.states[(int64_t)op.taxon * P.ncols_total + col] & 15u;
            mask = mask ? mask : 15u;
            const double* w = wtab + mask * 4;
            double wv[4] = {w[0], w[1], w[2], w[3]};
            Partial m;
            tip_message(M, wv, op.t * s, m);
            if (op.code == OP_TIP_SET) acc = m;
            else { partial_rescale(acc, scale); partial_mul(acc, m); }
        } else if (op.code == OP_BRANCH) {
            partial_rescale(acc, scale);
            branch_apply(M, op.t * s, acc);
        } else if (op.code == OP_PUSH) {
            double* slot = stack + (size_t)sp * 12 * kSiteBlock + lane;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                slot[(i)*kSiteBlock] = acc.v[i];
                slot[(4 + i) * kSiteBlock] = acc.d1[i];
                slot[(8 + i) * kSiteBlock] = acc.d2[i];
            }
            ++sp;
        } else {  // OP_POP_MUL
            --sp;
            const double* slot = stack + (size_t)sp * 12 * kSiteBlock + lane;
            Partial m;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                m.v[i] = slot[(i)*kSiteBlock];
                m.d1[i] = slot[(4 + i) * kSiteBlock];
                m.d2[i] = slot[(8 + i) * kSiteBlock];
            }
            partial_mul(acc, m);
        }
    }
    double L = 0, L1 = 0, L2 = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        L = fma(M->pi[i], acc.v[i], L);
        L1 = fma(M->pi[i], acc.d1[i], L1);
        L2 = fma(M->pi[i], acc.d2[i], L2);
    }
    double inv = 1.0 / L;
    g = L1 * inv;
    h = L2 * inv - g * g;
    f = log(L) + (double)scale * 0.6931471805599453;
}

__global__ __launch_bounds__(kSiteBlock) void site_rate_kernel(SiteParams P) {
    extern __shared__ double lds[];
    double* wtab = lds;          // [16 masks][4]
    double* stack = lds + 64;    // [stack_depth][12][64]
    const int chunk = blockIdx.x;
    const int locus = P.chunk_locus[chunk];
    const int cidx = P.chunk_index[chunk];
    const int count = P.work_count[locus];
    if (cidx * kSiteBlock >= count) return;  // this chunk's columns were answered by classify_kernel
    const LocusModel* __restrict__ M = P.models + locus;
    const int lane = threadIdx.x;
    {   // tip table: wtab[mask][k] = sum_{j in mask} Ui[k][j]
        int mask = lane >> 2, k = lane & 3;
        double w = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) w += ((mask >> j) & 1) ? M->Ui[k * 4 + j] : 0.0;
        wtab[lane] = w;
    }
    __syncthreads();
    const int64_t loff = P.locus_offsets[locus];
    const int widx = cidx * kSiteBlock + lane;
    const bool active = widx < count;
    const int64_t col = P.work_cols[loff + (active ? widx : cidx * kSiteBlock)];

    double u = 0.0, lo = kUMin, hi = kUMax, f = 0.0;
    bool lo_open = true, hi_open = true, done = !active;
    uint8_t flg = TPHIP_FLAG_MAXIT;
    unsigned evals = 0;
    for (int it = 0; it < kMaxIt; ++it) {
        if (__all(done)) break;
        double fe, g, h;
        evaluate_column(P, M, wtab, stack, col, exp(u), fe, g, h);
        if (!done) {
            ++evals;
            f = fe;
            const bool uphill = !(g <= 0.0);
            // Saturation: beyond this point g is second-order small under first-order rounding noise, its
            // sign is meaningless; report the policy value s = 1e4 (same rule as the oracle).
            if (fabs(g) < kFlatEps && fabs(h) < kFlatEps) { flg = TPHIP_FLAG_SATURATED; u = kUMax; done = true; }
            else if (u >= kUMax && uphill) { flg = TPHIP_FLAG_SATURATED; done = true; }
            else if (u <= kUMin && !uphill) { flg = TPHIP_FLAG_ZERO; done = true; }
            else {
                if (uphill) { lo = u; lo_open = false; } else { hi = u; hi_open = false; }
                double step = (h < 0.0) ? -g / h : (uphill ? kStepMax : -kStepMax);
                if (!(step <= kStepMax)) step = kStepMax;
                if (step < -kStepMax) step = -kStepMax;
                double un = u + step;
                // the bracket safeguard must not see a converged (possibly underflowing) Newton step
                if (fabs(step) >= kStepTol) {
                    if (un >= hi) un = hi_open ? kUMax : 0.5 * (lo + hi);
                    else if (un <= lo) un = lo_open ? kUMin : 0.5 * (lo + hi);
                    step = un - u;
                }
                if (fabs(step) < kStepTol) {
                    f = fma(step, fma(0.5 * h, step, g), f);  // f + g*step + h*step^2/2
                    flg = TPHIP_FLAG_OK;
                    done = true;
                }
                u = un;
            }
        }
    }
    if (active) {
        double s = exp(u);
        double r = s * M->kappa;
        P.rate[col] = r;
        P.subst[col] = r * P.chrono_length;
        P.lnl[col] = f;
        P.flag[col] = flg;
    }
    // evaluation count for the FLOP model (one atomic per wave)
    unsigned tot = evals;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) tot += __shfl_xor(tot, o);
    if (lane == 0 && P.eval_counter) atomicAdd(P.eval_counter, (unsigned long long)tot);
}

// Diagnostic: evaluate f = log L, g = df/du, h = d2f/du2 at a caller-chosen u for EVERY column (no
// classification, no optimiser).  Tests use it to check the derivative propagation by finite differences.
struct EvalParams {
    SiteParams S;
    const double* u;
    double* f;
    double* g;
    double* h;
};

__global__ __launch_bounds__(kSiteBlock) void eval_columns_kernel(EvalParams E) {
    extern __shared__ double lds[];
    double* wtab = lds;
    double* stack = lds + 64;
    const SiteParams& P = E.S;
    const int chunk = blockIdx.x;
    const int locus = P.chunk_locus[chunk];
    const int cidx = P.chunk_index[chunk];
    const LocusModel* __restrict__ M = P.models + locus;
    const int lane = threadIdx.x;
    {
        int mask = lane >> 2, k = lane & 3;
        double w = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) w += ((mask >> j) & 1) ? M->Ui[k * 4 + j] : 0.0;
        wtab[lane] = w;
    }
    __syncthreads();
    const int64_t lo = P.locus_offsets[locus], hi = P.locus_offsets[locus + 1];
    const int64_t want = lo + (int64_t)cidx * kSiteBlock + lane;
    const bool active = want < hi;
    const int64_t col = active ? want : lo;
    double f, g, h;
    evaluate_column(P, M, wtab, stack, col, exp(E.u[col]), f, g, h);
    if (active) { E.f[col] = f; E.g[col] = g; E.h[col] = h; }
}

}  // namespace tphip
