// stage1_opt_kernels.hpp -- the optimisers of HyPhy's stage 1 as device kernels.
//
// What they replace: every `Optimize(lf_MLES, lf)` of tapir/data/models_and_rates.bf -- the general reversible model with
// free branch lengths (bf:487-520) and the 202 rate-class models on the stashed lengths (bf:542-661) -- plus the Akaike
// averaging of bf:806-847.  The objective and its gradient are the likelihood kernels (locus_value_kernel.hpp,
// locus_lik_kernel.hpp); everything AROUND them -- search direction, line search bookkeeping, quasi-Newton updates, stopping,
// boundary escapes, model screening and pruning -- is in this file, one problem per wave (general model: 5 + 2N-3
// coordinates) or per thread (rate-class models: at most 4), with the optimiser state of all problems in HBM.  The host
// (stage1_driver.hip) only sequences launches and reads a handful of counters per step.
//
// A problem moves through phases inside one iteration:
//   LIVE -> direction kernel -> PEND (a trial step is due) | RESTART (moved off a boundary trap) | CONV
//   PEND -> trial kernel emits a candidate -> value kernel -> accept kernel -> ACCEPTED | PEND (shorter step) | FAILED
//   ACCEPTED / RESTART -> point kernel emits a candidate -> gradient kernel (or difference stencil) -> update kernel -> LIVE
//   FAILED -> update kernel: forget the metric and retry once, else CONV
// Candidates are appended to flat arrays through an atomic cursor; a candidate's value does not depend on its position.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include "gtr_model.hpp"

namespace tphip {
namespace s1 {

// bounds and rules of the search (same numbers as round 2's host optimiser)
constexpr double kLogRateMin = -7.0, kLogRateMax = 9.2;     // exchangeabilities within [9e-4, 1e4] (HyPhy: [0, 10000])
constexpr double kLogBlenMin = -23.0, kLogBlenMax = 4.0;    // branch lengths within [1e-10, 55]
constexpr double kMaxLogStep = 2.0;                         // largest move of a log-parameter in one iteration
constexpr double kEscapeRate = 0.05, kEscapeLength = 1e-3;  // where a wrongly collapsed rate / branch is put back
constexpr double kPruneNats = 21.0;                         // a model this far behind weighs < e^-21 = 8e-10
constexpr double kFtol = 1e-10, kGtol = 2e-6, kPtol = 1e-9;
constexpr int kHistory = 8;                                 // L-BFGS correction pairs
constexpr int kMaxLineSearch = 30;
constexpr int kModels = 203, kSubModels = 202;
constexpr int kScreenPoints = 31;
constexpr double kScreenStep = 2e-2;

enum : int32_t { C_NCAND = 0, C_NLIVE = 1, C_NPEND = 2, C_NGRAD = 3, C_NFAILED = 4, C_NPRUNED = 5, C_NKEEP = 6, C_PARS = 7, C_COUNT = 8 };
enum : uint8_t { PH_LIVE = 0, PH_PEND = 1, PH_ACCEPTED = 2, PH_FAILED = 3, PH_RESTART = 4, PH_CONV = 5, PH_INIT = 6 };

// flat candidate arrays in the layout tphip_locus_loglik_dev / tphip_locus_gradient_dev take
struct CandArrays {
    int32_t* locus;   // [cap]
    double* exch;     // [cap][6]
    int32_t* vec;     // [cap]
    double* scale;    // [cap]
    int32_t* pidx;    // [cap] always -1 here
    double* pfac;     // [cap] always 1
    int32_t* prob;    // [cap] problem the candidate belongs to
    double* out;      // [cap] log-likelihoods
};

__device__ inline double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__device__ inline double wave_max(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o));
    return v;
}
__device__ inline void atomic_max_double(double* addr, double v) {
    unsigned long long* a = (unsigned long long*)addr;
    unsigned long long old = *a;
    while (__longlong_as_double(old) < v) {
        const unsigned long long seen = atomicCAS(a, old, (unsigned long long)__double_as_longlong(v));
        if (seen == old) break;
        old = seen;
    }
}
// next trial step after `t` failed the Armijo test: minimiser of the parabola through f(0) = f0, f'(0) = gd < 0 and f(t) = ft,
// kept within [0.1 t, 0.5 t] (0.25 t where the trial value is not finite)
__device__ inline double backtrack(double t, double ft, double f0, double gd) {
    const double den = 2.0 * (ft - f0 - gd * t);
    const bool good = isfinite(ft) && den > 0;
    const double tq = good ? -gd * t * t / den : 0.25 * t;
    return fmax(fmin(tq, 0.5 * t), 0.1 * t);
}

// ---------------------------------------------------------------------------------------------------------------------
// General reversible model (bf:487-520): coordinates (log AC, log AT, log CG, log CT, log GT, log b_1 .. log b_nb) with
// b = t * totalFactor(r) the branch lengths in expected substitutions; batched L-BFGS, diagonal initial metric from the
// gradient kernel's curvature of the branch lengths, Armijo backtracking.
// ---------------------------------------------------------------------------------------------------------------------
struct GrmState {
    int32_t P, D, nb, nn;
    int32_t bspace_metric;       // 1: initial metric of a branch coordinate from the Newton step of b itself (grm_direction_kernel)
    // Branch coordinates.  On a reversible model the likelihood depends on the two branches below a bifurcating root only
    // through their SUM (P(b1) P(b2) = P(b1 + b2) across the root): left as two coordinates that is an exactly flat, in
    // log-coordinates curved, direction, which made quasi-Newton steps overshoot for hundreds of iterations on some loci.  The
    // two are therefore ONE coordinate s with b1 = w1 s, b2 = w2 s at the input tree's proportion (w1 + w2 = 1).
    const int32_t* node_coord;   // [nn] coordinate (0-based among the branch coordinates) of the node's branch, -1 for the root
    const double* node_w;        // [nn] the node's share of its coordinate (1 except for the two branches below the root)
    const int32_t* branches;     // [nb] node of branch coordinate j (the first of the two for the root pair)
    const int32_t* partner;      // [nb] second node of the coordinate, -1 for none
    const double* dk;            // [P][6] 2 pi_i pi_j: totalFactor = exch . dk
    double *x, *g, *hd, *d, *xt; // [P][D]
    double *S, *Y;               // [kHistory][P][D]
    double* rho;                 // [kHistory][P]
    double *f, *fnew, *t, *gd, *gamma, *last_df;   // [P]
    int32_t *nhist, *head, *kicks, *iters, *ls_round, *slot;   // [P]
    uint8_t *phase, *fresh;      // [P]
    int32_t* counters;
    CandArrays C;
    double* vecs;                // [cap][nn] branch-length vectors of the candidates (cap = P)
    double *o_lnl, *o_dexch, *o_dlogt, *o_sdl, *o_d2;   // gradient kernel outputs per candidate
};

__device__ inline int grm_free_of(int q) { return q == 0 ? 0 : q - 1; }   // rate q of (AC, AG, AT, CG, CT, GT) -> free coordinate (q != 1)

// candidate `slot` <- point X of problem p (one wave)
__device__ inline void grm_emit_point(const GrmState& G, int p, const double* X, int slot) {
    const int lane = threadIdx.x;
    double e = 0.0;
    if (lane < 6) e = lane == 1 ? 1.0 : exp(X[grm_free_of(lane)]);
    const double tf = wave_sum(lane < 6 ? e * G.dk[(size_t)p * 6 + lane] : 0.0);
    if (lane < 6) G.C.exch[(size_t)slot * 6 + lane] = e;
    double* v = G.vecs + (size_t)slot * G.nn;
    for (int n = lane; n < G.nn; n += 64) {
        const int j = G.node_coord[n];
        v[n] = j >= 0 ? exp(X[5 + j]) * G.node_w[n] : 0.0;
    }
    if (lane == 0) {
        G.C.locus[slot] = p; G.C.vec[slot] = slot; G.C.scale[slot] = 1.0 / tf; G.C.pidx[slot] = -1; G.C.pfac[slot] = 1.0;
        G.C.prob[slot] = p;
    }
}

// search direction, stopping test and boundary escape of every LIVE problem; block = one wave = one problem
__global__ __launch_bounds__(64) void grm_direction_kernel(GrmState G) {
    extern __shared__ double q[];   // [D]
    const int p = blockIdx.x, lane = threadIdx.x;
    if (G.phase[p] != PH_LIVE) return;
    const int D = G.D, P = G.P;
    const double* g = G.g + (size_t)p * D;
    const double* hd = G.hd + (size_t)p * D;
    const double* xc = G.x + (size_t)p * D;
    // Projected gradient: a coordinate AT a bound whose gradient points out of the box is fixed for this iteration -- it takes
    // no part in the direction, in the predicted decrease or in the stopping test.  (Without this a conserved locus, most of
    // whose branches end at the lower bound, never met the stopping test: dozens of bound coordinates each added -2 g_j to
    // the "predicted decrease", the line search shrank to 1e-12 every iteration, 300 iterations of 23 likelihood calls.)
    auto fixed = [&](int j) {
        const double lo = j < 5 ? kLogRateMin : kLogBlenMin, hi = j < 5 ? kLogRateMax : kLogBlenMax;
        return (xc[j] <= lo && g[j] > 0.0) || (xc[j] >= hi && g[j] < 0.0);
    };
    for (int j = lane; j < D; j += 64) q[j] = fixed(j) ? 0.0 : g[j];
    const int head = G.head[p];
    const int cnt = head < kHistory ? head : kHistory;
    double alpha[kHistory];
    // two-loop recursion, newest pair first (rho = 0 marks a slot without a usable pair)
#pragma unroll
    for (int jj = 0; jj < kHistory; ++jj) {
        alpha[jj] = 0.0;
        if (jj < cnt) {
            const int i = (head - 1 - jj) % kHistory;
            const double* Si = G.S + ((size_t)i * P + p) * D;
            const double* Yi = G.Y + ((size_t)i * P + p) * D;
            const double rho = G.rho[(size_t)i * P + p];
            double part = 0.0;
            for (int j = lane; j < D; j += 64) part = fma(Si[j], q[j], part);
            const double a = rho * wave_sum(part);
            alpha[jj] = a;
            for (int j = lane; j < D; j += 64) q[j] = fixed(j) ? 0.0 : fma(-a, Yi[j], q[j]);
        }
    }
    // initial inverse metric: 1 / curvature where the objective supplies a positive one, the secant scalar elsewhere
    const double gamma = G.gamma[p];
    bool any_h = false;
    for (int j = lane; j < D; j += 64) {
        double h = hd[j];
        any_h |= isfinite(h);
        if (G.bspace_metric && j >= 5 && isfinite(h)) {
            // The curvature in log b is a poor model of a short branch: there lnL is close to LINEAR in b, so in log b the
            // second derivative h = b^2 f'' + b f' nearly cancels against the first (g = b f') and a Newton step -g / h is
            // tens of log-units (seen: g = -0.09, h = 9e-4), clipped, and the line search then shortens EVERY coordinate's
            // step.  In b itself f is convex there: take the Newton step of b, r = db / b = -g / (h - g), express it as a
            // step of log b, log(1 + r), and use the curvature that step implies.  A branch on its way to zero (r <= -0.9:
            // the step in b overshoots zero) moves by log 0.1 per iteration.
            const double gj = g[j], c = h - gj;   // b^2 f''
            if (c > 1e-300 && gj != 0.0) {
                const double r = -gj / c;
                const double step = r > -0.9 ? log1p(r) : -2.302585092994046;
                h = -gj / step;
            }
        }
        const bool okh = isfinite(h) && h > 1e-12;
        q[j] *= okh ? 1.0 / h : gamma;
    }
    const bool have_h = __any(any_h);
#pragma unroll
    for (int jj = kHistory - 1; jj >= 0; --jj) {
        if (jj < cnt) {
            const int i = (head - 1 - jj) % kHistory;
            const double* Si = G.S + ((size_t)i * P + p) * D;
            const double* Yi = G.Y + ((size_t)i * P + p) * D;
            const double rho = G.rho[(size_t)i * P + p];
            double part = 0.0;
            for (int j = lane; j < D; j += 64) part = fma(Yi[j], q[j], part);
            const double b = rho * wave_sum(part);
            for (int j = lane; j < D; j += 64) q[j] = fixed(j) ? 0.0 : fma(alpha[jj] - b, Si[j], q[j]);
        }
    }
    // d = clip(-r) per coordinate: one runaway parameter must not shrink the others' step
    double gdp = 0.0, ggp = 0.0, gmaxp = 0.0;
    for (int j = lane; j < D; j += 64) {
        const bool fx = fixed(j);
        const double gj = fx ? 0.0 : g[j];
        const double dj = fx ? 0.0 : fmax(fmin(-q[j], kMaxLogStep), -kMaxLogStep);
        q[j] = dj;
        gdp = fma(gj, dj, gdp);
        ggp = fma(gj, gj, ggp);
        gmaxp = fmax(gmaxp, fabs(gj));
    }
    double gd = wave_sum(gdp);
    const double gg = wave_sum(ggp), gmax = wave_max(gmaxp);
    if (!(gd < 0)) {
        for (int j = lane; j < D; j += 64) q[j] = fixed(j) ? 0.0 : -g[j];
        gd = -gg;
    }
    double dmaxp = 0.0;
    for (int j = lane; j < D; j += 64) dmaxp = fmax(dmaxp, fabs(q[j]));
    const double dmax = wave_max(dmaxp);
    const double f = G.f[p], scale_f = 1.0 + fabs(f), last_df = G.last_df[p];
    // stop: the last step gained (almost) nothing AND the quasi-Newton model predicts that the next one will not either
    const bool stop = (last_df <= kFtol * scale_f && -gd <= kPtol * scale_f && gmax <= kGtol * scale_f) || gmax <= 1e-9;
    if (stop) {
        // Log-parameters have a trap at zero: d f / d log b = b * d f / d b vanishes with b whatever d f / d b is.  At a would-be
        // stopping point every collapsed branch (and every rate at its lower bound) is tested in the ORIGINAL parametrisation
        // (optimality at the bound needs d f / d b >= 0); offenders are put back at a small positive value.
        double* x = G.x + (size_t)p * D;
        const bool few = G.kicks[p] < 3;
        bool moved = false;
        for (int j = lane; j < D; j += 64) {
            const double X = x[j], v = exp(X), slope = g[j] / v;
            if (j < 5) {
                if (X < kLogRateMin + 1.0 && slope * kEscapeRate < -1e-7 * scale_f && few) { x[j] = log(kEscapeRate); moved = true; }
            } else {
                if (v < 1e-6 && slope * kEscapeLength < -1e-7 * scale_f && few) { x[j] = log(kEscapeLength); moved = true; }
            }
        }
        moved = __any(moved);
        if (moved) {
            if (lane < kHistory) G.rho[(size_t)lane * P + p] = 0.0;
            if (lane == 0) {
                G.kicks[p] += 1; G.nhist[p] = 0; G.last_df[p] = INFINITY; G.phase[p] = PH_RESTART;
                atomicAdd(&G.counters[C_NGRAD], 1);
            }
        } else if (lane == 0) {
            G.phase[p] = PH_CONV;
            atomicSub(&G.counters[C_NLIVE], 1);
        }
        return;
    }
    double* d = G.d + (size_t)p * D;
    for (int j = lane; j < D; j += 64) d[j] = q[j];
    if (lane == 0) {
        const double step0 = (G.nhist[p] == 0 && !have_h) ? fmin(1.0, 1.0 / fmax(gmax, 1e-300)) : 1.0;
        G.t[p] = fmin(step0, kMaxLogStep / fmax(dmax, 1e-300));
        G.gd[p] = gd;
        G.ls_round[p] = 0;
        G.phase[p] = PH_PEND;
        atomicAdd(&G.counters[C_NPEND], 1);
    }
}

// trial point x + t d of every PEND problem -> value candidate
__global__ __launch_bounds__(64) void grm_trial_kernel(GrmState G) {
    const int p = blockIdx.x, lane = threadIdx.x;
    if (G.phase[p] != PH_PEND) return;
    const int D = G.D;
    extern __shared__ double xs[];   // [D] the trial point, for the lanes that read other lanes' coordinates
    const double* x = G.x + (size_t)p * D;
    const double* d = G.d + (size_t)p * D;
    double* xt = G.xt + (size_t)p * D;
    const double t = G.t[p];
    for (int j = lane; j < D; j += 64) {
        const double lo = j < 5 ? kLogRateMin : kLogBlenMin, hi = j < 5 ? kLogRateMax : kLogBlenMax;
        const double v = fmax(fmin(fma(t, d[j], x[j]), hi), lo);
        xt[j] = v;
        xs[j] = v;
    }
    int slot = 0;
    if (lane == 0) slot = atomicAdd(&G.counters[C_NCAND], 1);
    slot = __shfl(slot, 0);
    if (lane == 0) G.slot[p] = slot;
    __syncthreads();
    grm_emit_point(G, p, xs, slot);
}

// Armijo test of the trial values; thread = candidate
__global__ __launch_bounds__(256) void grm_accept_kernel(GrmState G, int ncand) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= ncand) return;
    const int p = G.C.prob[c];
    const double ft = -G.C.out[c], f = G.f[p], t = G.t[p], gd = G.gd[p];
    if (isfinite(ft) && ft <= f + 1e-4 * t * gd) {
        G.fnew[p] = ft;
        G.phase[p] = PH_ACCEPTED;
        atomicAdd(&G.counters[C_NGRAD], 1);
    } else if (++G.ls_round[p] >= kMaxLineSearch) {
        G.phase[p] = PH_FAILED;   // no decrease found along d
        atomicAdd(&G.counters[C_NFAILED], 1);
    } else {
        G.t[p] = backtrack(t, ft, f, gd);
        atomicAdd(&G.counters[C_NPEND], 1);
    }
}

// gradient candidates: the accepted point of ACCEPTED problems, the moved point of RESTART problems (and every problem at start)
__global__ __launch_bounds__(64) void grm_point_kernel(GrmState G) {
    const int p = blockIdx.x, lane = threadIdx.x;
    const uint8_t ph = G.phase[p];
    if (ph != PH_ACCEPTED && ph != PH_RESTART) return;
    const double* X = (ph == PH_ACCEPTED ? G.xt : G.x) + (size_t)p * G.D;
    int slot = 0;
    if (lane == 0) slot = atomicAdd(&G.counters[C_NCAND], 1);
    slot = __shfl(slot, 0);
    if (lane == 0) G.slot[p] = slot;
    grm_emit_point(G, p, X, slot);
}

// absorb the gradient kernel's outputs: chain rule into (log r, log b), correction pair, bookkeeping
__global__ __launch_bounds__(64) void grm_update_kernel(GrmState G) {
    const int p = blockIdx.x, lane = threadIdx.x;
    const uint8_t ph = G.phase[p];
    const int D = G.D, P = G.P;
    if (ph == PH_FAILED) {
        // a failed line search with correction pairs in play: forget them and try once more along the preconditioned
        // gradient before giving the point up as converged
        const bool retry = G.nhist[p] > 0;
        if (retry && lane < kHistory) G.rho[(size_t)lane * P + p] = 0.0;
        if (lane == 0) {
            G.iters[p] += 1;
            if (retry) { G.nhist[p] = 0; G.last_df[p] = INFINITY; G.phase[p] = PH_LIVE; }
            else { G.last_df[p] = 0.0; G.phase[p] = PH_CONV; atomicSub(&G.counters[C_NLIVE], 1); }
        }
        return;
    }
    if (ph != PH_ACCEPTED && ph != PH_RESTART) return;
    const int c = G.slot[p];
    const double fx = -G.o_lnl[c], scale = G.C.scale[c], sdl = G.o_sdl[c];
    double* x = G.x + (size_t)p * D;
    double* g = G.g + (size_t)p * D;
    double* hd = G.hd + (size_t)p * D;
    const double* xt = G.xt + (size_t)p * D;
    const int slot_i = G.head[p] % kHistory;
    double* Si = G.S + ((size_t)slot_i * P + p) * D;
    double* Yi = G.Y + ((size_t)slot_i * P + p) * D;
    double syp = 0.0, yyp = 0.0, ssp = 0.0;
    for (int j = lane; j < D; j += 64) {
        double gx, hx;
        if (j < 5) {
            // t_b = b_b / totalFactor(r): d log t_b / d r_q = -(2 pi_i pi_j) / totalFactor for every branch
            const int qq = j == 0 ? 0 : j + 1;
            const double e = G.C.exch[(size_t)c * 6 + qq];
            gx = -(G.o_dexch[(size_t)c * 6 + qq] - sdl * G.dk[(size_t)p * 6 + qq] * scale) * e;
            hx = NAN;
        } else {
            const int n = G.branches[j - 5], n2 = G.partner[j - 5];
            const double g1 = G.o_dlogt[(size_t)c * G.nn + n], h1 = G.o_d2[(size_t)c * G.nn + n];
            if (n2 < 0) {
                gx = -g1;
                hx = -h1;
            } else {
                // lnL = phi(b1 + b2): d/d log s = g1 + g2; with phi' = g1 / b1 and phi'' = (h1 - g1) / b1^2 (b1 = w1 s),
                // d2/d(log s)^2 = s^2 phi'' + s phi' = (h1 - g1) / w1^2 + g1 / w1
                const double w1 = G.node_w[n];
                gx = -(g1 + G.o_dlogt[(size_t)c * G.nn + n2]);
                hx = -((h1 - g1) / (w1 * w1) + g1 / w1);
            }
        }
        if (ph == PH_ACCEPTED) {
            const double s = xt[j] - x[j], y = gx - g[j];
            Si[j] = s; Yi[j] = y;
            syp = fma(s, y, syp); yyp = fma(y, y, yyp); ssp = fma(s, s, ssp);
            x[j] = xt[j];
        }
        g[j] = gx;
        hd[j] = hx;
    }
    if (ph == PH_RESTART) {
        if (lane == 0) { G.f[p] = fx; G.phase[p] = PH_LIVE; }
        return;
    }
    const double sy = wave_sum(syp), yy = wave_sum(yyp), ss = wave_sum(ssp);
    if (lane == 0) {
        const bool upd = sy > 1e-12 * sqrt(ss * yy + 1e-300);
        G.rho[(size_t)slot_i * P + p] = upd ? 1.0 / sy : 0.0;
        G.head[p] += 1;
        if (upd) { G.gamma[p] = sy / fmax(yy, 1e-300); G.nhist[p] += 1; }
        G.last_df[p] = G.f[p] - fx;
        G.f[p] = fx;
        G.iters[p] += 1;
        G.phase[p] = PH_LIVE;
    }
}

// HarvestFrequencies(Freqs, filter, 1, 1, 1) (bf:968) from the per-locus histogram of state masks: every cell adds
// 1 / popcount(mask) to each base it may be (a gap counts 1/4 to each); a locus without cells gets 1/4 each.  The floor
// keeps a base that never occurs from making the eigen-form singular (as at plan creation).  thread = locus
__global__ __launch_bounds__(64) void empirical_pi_kernel(int32_t L, const unsigned long long* hist, double* pi_out) {
    const int l = blockIdx.x * blockDim.x + threadIdx.x;
    if (l >= L) return;
    double c[4] = {0, 0, 0, 0};
    for (int m = 0; m < 16; ++m) {
        const double n = (double)hist[(size_t)l * 16 + m];
        const int mm = m ? m : 15;
        const double w = n / (double)__popc(mm);
        for (int k = 0; k < 4; ++k) if ((mm >> k) & 1) c[k] += w;
    }
    const double tot = c[0] + c[1] + c[2] + c[3];
    for (int k = 0; k < 4; ++k) pi_out[(size_t)l * 4 + k] = tot > 0 ? fmax(c[k] / tot, 1e-12) : 0.25;
}

// dk[l][q] = 2 pi_i pi_j over the rate pairs (AC, AG, AT, CG, CT, GT): totalFactor(r) = r . dk (bf:531-534); thread = locus
__global__ __launch_bounds__(64) void dk_kernel(int32_t L, const LocusModel* models, double* dk) {
    const int l = blockIdx.x * blockDim.x + threadIdx.x;
    if (l >= L) return;
    const double* pi = models[l].pi;
    const int PI_[6] = {0, 0, 0, 1, 1, 2}, PJ_[6] = {1, 2, 3, 2, 3, 3};
    for (int q = 0; q < 6; ++q) dk[(size_t)l * 6 + q] = 2.0 * pi[PI_[q]] * pi[PJ_[q]];
}

// start of the branch lengths: the input tree's shape rescaled to the best of a grid of scales under the all-ones model
__global__ __launch_bounds__(256) void grm_grid_emit_kernel(CandArrays C, int32_t P, int32_t ngrid, const double* grid) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)P * ngrid) return;
    const int p = (int)(i / ngrid), k = (int)(i % ngrid);
    C.locus[i] = p; C.vec[i] = 0; C.scale[i] = grid[k]; C.pidx[i] = -1; C.pfac[i] = 1.0; C.prob[i] = p;
#pragma unroll
    for (int q = 0; q < 6; ++q) C.exch[i * 6 + q] = 1.0;
}
__global__ __launch_bounds__(256) void grm_grid_pick_kernel(GrmState G, int32_t ngrid, const double* grid, const double* shape) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= G.P) return;
    int best = 0;
    double bv = G.C.out[(size_t)p * ngrid];
    for (int k = 1; k < ngrid; ++k) {
        const double v = G.C.out[(size_t)p * ngrid + k];
        if (v > bv) { bv = v; best = k; }
    }
    G.fnew[p] = bv;   // for the comparison with the parsimony start (grm_pars_pick_kernel)
    double tf = 0.0;
#pragma unroll
    for (int q = 0; q < 6; ++q) tf += G.dk[(size_t)p * 6 + q];
    double* x = G.x + (size_t)p * G.D;
#pragma unroll
    for (int j = 0; j < 5; ++j) x[j] = 0.0;
    for (int j = 0; j < G.nb; ++j) {
        const int n = G.branches[j], n2 = G.partner[j];
        x[5 + j] = log(grid[best] * (shape[n] + (n2 >= 0 ? shape[n2] : 0.0)) * tf);
    }
    for (int j = 0; j < G.D; ++j) G.d[(size_t)p * G.D + j] = x[j];   // the prior of the shrunk parsimony start
}

// ---- second start of the branch lengths: per-branch parsimony counts --------------------------------------
// changes[l][n] += weight of the columns of locus l whose most parsimonious reconstruction changes state on the branch above
// node n; changes[l][nn] += total weight.  Fitch's two passes per column over the post-order node numbering (children before
// parents, root last -- checked by the caller): down, the state sets (tips: their 4-bit masks, a gap = all four); up, the
// parent's state where the child's set allows it, else the child's lowest state and one change.  thread = column, a block's
// counts gathered in LDS, one global atomic per (block, branch).
constexpr int kParsMaxNodes = 1024;
__global__ __launch_bounds__(128) void branch_parsimony_kernel(const uint8_t* states, int64_t ncols_total, const int64_t* locus_offsets,
                                                               const double* col_weight, int32_t nn, const int32_t* parent,
                                                               const int32_t* leaf_taxon, double* changes) {
    extern __shared__ double acc[];   // [nn + 1]
    const int l = blockIdx.x;
    for (int n = threadIdx.x; n <= nn; n += blockDim.x) acc[n] = 0.0;
    __syncthreads();
    const int64_t lo = locus_offsets[l], hi = locus_offsets[l + 1];
    for (int64_t col = lo + (int64_t)blockIdx.y * blockDim.x + threadIdx.x; col < hi; col += (int64_t)gridDim.y * blockDim.x) {
        uint8_t set[kParsMaxNodes];
        for (int n = 0; n < nn; ++n) {
            const int t = leaf_taxon[n];
            unsigned m = 0;
            if (t >= 0) { m = states[(int64_t)t * ncols_total + col] & 15u; m = m ? m : 15u; }
            set[n] = (uint8_t)m;
        }
        for (int n = 0; n + 1 < nn; ++n) {
            const int pa = parent[n];
            const unsigned a = set[pa], c = set[n], both = a & c;
            set[pa] = (uint8_t)(a == 0 ? c : (both ? both : (a | c)));
        }
        const double w = col_weight ? col_weight[col] : 1.0;
        { const unsigned r = set[nn - 1]; set[nn - 1] = (uint8_t)(r & (0u - r)); }
        for (int n = nn - 2; n >= 0; --n) {
            const unsigned ps = set[parent[n]], c = set[n];
            if (c & ps) set[n] = (uint8_t)ps;
            else {
                set[n] = (uint8_t)(c & (0u - c));
                if (c) atomicAdd(&acc[n], w);
            }
        }
        atomicAdd(&acc[nn], w);
    }
    __syncthreads();
    for (int n = threadIdx.x; n <= nn; n += blockDim.x)
        if (acc[n] != 0.0) atomicAdd(&changes[(size_t)l * (nn + 1) + n], acc[n]);
}
// the parsimony point of every problem: rates 1, b = -3/4 log(1 - 4/3 p) with p = changes per column on the branch (at least
// kParsFloor changes, at most 0.6 per column) -> G.xt and value candidate p.  block = one wave = one problem
constexpr double kParsFloor = 0.3;
// alpha >= 0: the counts shrunk towards the grid start b0 (kept in G.d by grm_grid_pick_kernel) as a Gamma prior of alpha
// pseudo-changes: b = (b_pars W + alpha) / (W + alpha / b0) -- branches with many changes follow the data, branches with
// hardly any follow the input tree's shape
__global__ __launch_bounds__(64) void grm_pars_emit_kernel(GrmState G, const double* changes, double alpha) {
    extern __shared__ double xs[];   // [D]
    const int p = blockIdx.x, lane = threadIdx.x;
    const double* ch = changes + (size_t)p * (G.nn + 1);
    const double W = fmax(ch[G.nn], 1.0);
    double* xt = G.xt + (size_t)p * G.D;
    for (int j = lane; j < G.D; j += 64) {
        double v = 0.0;
        if (j >= 5) {
            const int n = G.branches[j - 5], n2 = G.partner[j - 5];
            const double c = ch[n] + (n2 >= 0 ? ch[n2] : 0.0);
            const double pc = fmin(fmax(c, alpha >= 0.0 ? 0.0 : kParsFloor) / W, 0.6);
            double b = -0.75 * log1p(-pc * (4.0 / 3.0));
            if (alpha >= 0.0) {
                const double b0 = exp(G.d[(size_t)p * G.D + j]);
                b = (b * W + alpha) / (W + alpha / b0);
            }
            v = fmax(fmin(log(b), kLogBlenMax), kLogBlenMin);
        }
        xt[j] = v;
        xs[j] = v;
    }
    __syncthreads();
    grm_emit_point(G, p, xs, p);
}
// keep the better of the two starts (G.fnew holds the grid start's log-likelihood, candidate p the parsimony point's)
__global__ __launch_bounds__(64) void grm_pars_pick_kernel(GrmState G, int32_t* taken, double margin) {
    const int p = blockIdx.x, lane = threadIdx.x;
    const double v = G.C.out[p];
    if (!(isfinite(v) && v > G.fnew[p] + margin)) return;
    for (int j = lane; j < G.D; j += 64) G.x[(size_t)p * G.D + j] = G.xt[(size_t)p * G.D + j];
    if (lane == 0) G.fnew[p] = v;
    if (lane == 0 && taken) atomicAdd(taken, 1);
}

// fitted general model -> exchangeabilities [P][6], branch lengths t = b / totalFactor [P][nn], stash b [P][nn], lnL [P]
__global__ __launch_bounds__(64) void grm_finish_kernel(GrmState G, double* exch, double* blen_t, double* stash, double* lnl) {
    const int p = blockIdx.x, lane = threadIdx.x;
    const double* X = G.x + (size_t)p * G.D;
    double e = 0.0;
    if (lane < 6) e = lane == 1 ? 1.0 : exp(X[grm_free_of(lane)]);
    const double tf = wave_sum(lane < 6 ? e * G.dk[(size_t)p * 6 + lane] : 0.0);
    if (lane < 6) exch[(size_t)p * 6 + lane] = e;
    for (int n = lane; n < G.nn; n += 64) {
        const int j = G.node_coord[n];
        const double b = j >= 0 ? exp(X[5 + j]) * G.node_w[n] : 0.0;
        stash[(size_t)p * G.nn + n] = b;
        blen_t[(size_t)p * G.nn + n] = b / tf;
    }
    if (lane == 0) lnl[p] = -G.f[p];
}

// ---------------------------------------------------------------------------------------------------------------------
// Quadratic screen of the 202 rate-class models (see stage1_driver.hip): every model maximises the SAME function
// f(rho) = lnL at the stashed lengths as a function of the five log-rates, over a linear subspace rho = A_m theta.
// ---------------------------------------------------------------------------------------------------------------------
struct ScreenState {
    int32_t L;
    const double* grm_exch;   // [L][6]
    const double* grm_lnl;    // [L]
    const double* w6;         // [L][6]
    const int8_t* cls;        // [202][6] class of each rate, -1 = the class of AG (fixed at 1), free classes 0..k-1
    const int32_t* kk;        // [202]
    double* K;                // [L][25] curvature of -lnL in the five log-rates, made safely positive definite
    double* g5;               // [L][5]  d lnL / d rho at the general model's optimum
    double* theta;            // [L*202][4] screened optimum of every model
    double* hdiag;            // [L*202][4] its curvature per class rate (NaN beyond k)
    double* f_at;             // [L*202] -lnL at theta
    double* best;             // [L] best Akaike score lnL - k seen so far
    int32_t* keep;            // [L*202] compacted list of surviving problems
    int32_t* counters;
    CandArrays C;
};

// offsets of the 31-point stencil: 0, +-h e_i, +-h (e_i + e_j)
__device__ inline void screen_point(int pt, double h, double* e) {
#pragma unroll
    for (int i = 0; i < 5; ++i) e[i] = 0.0;
    if (pt == 0) return;
    if (pt <= 10) { e[(pt - 1) >> 1] = ((pt - 1) & 1) ? -h : h; return; }
    const int qd = (pt - 11) >> 1;
    const double s = ((pt - 11) & 1) ? -h : h;
    int i = 0, j = 1, c = 0;
    for (int a = 0; a < 5; ++a)
        for (int b = a + 1; b < 5; ++b) { if (c == qd) { i = a; j = b; } ++c; }
    e[i] = s; e[j] = s;
}

__global__ __launch_bounds__(256) void screen_emit_kernel(ScreenState S) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)S.L * kScreenPoints) return;
    const int l = (int)(i / kScreenPoints), pt = (int)(i % kScreenPoints);
    double e[5];
    screen_point(pt, kScreenStep, e);
    double tf = 0.0;
#pragma unroll
    for (int q = 0; q < 6; ++q) {
        const double r = q == 1 ? 1.0 : S.grm_exch[(size_t)l * 6 + q] * exp(e[grm_free_of(q)]);
        S.C.exch[i * 6 + q] = r;
        tf = fma(r, S.w6[(size_t)l * 6 + q], tf);
    }
    S.C.locus[i] = l; S.C.vec[i] = l; S.C.scale[i] = 1.0 / tf; S.C.pidx[i] = -1; S.C.pfac[i] = 1.0; S.C.prob[i] = l;
}

// cyclic Jacobi for a symmetric 5x5 matrix A (destroyed): eigenvalues w, eigenvectors as columns of V
__device__ inline void jacobi5(double A[5][5], double V[5][5], double w[5]) {
    for (int i = 0; i < 5; ++i) for (int j = 0; j < 5; ++j) V[i][j] = (i == j) ? 1.0 : 0.0;
    for (int sweep = 0; sweep < 40; ++sweep) {
        double off = 0, dia = 0;
        for (int p = 0; p < 5; ++p) { dia += A[p][p] * A[p][p]; for (int q2 = p + 1; q2 < 5; ++q2) off += A[p][q2] * A[p][q2]; }
        if (off <= 1e-30 * dia || off < 1e-290) break;
        for (int p = 0; p < 5; ++p) for (int q2 = p + 1; q2 < 5; ++q2) {
            const double apq = A[p][q2];
            if (apq == 0.0) continue;
            const double th = (A[q2][q2] - A[p][p]) / (2.0 * apq);
            const double t = (th >= 0 ? 1.0 : -1.0) / (fabs(th) + sqrt(th * th + 1.0));
            const double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
            for (int k = 0; k < 5; ++k) { const double x = A[k][p], y = A[k][q2]; A[k][p] = c * x - s * y; A[k][q2] = s * x + c * y; }
            for (int k = 0; k < 5; ++k) { const double x = A[p][k], y = A[q2][k]; A[p][k] = c * x - s * y; A[q2][k] = s * x + c * y; }
            for (int k = 0; k < 5; ++k) { const double x = V[k][p], y = V[k][q2]; V[k][p] = c * x - s * y; V[k][q2] = s * x + c * y; }
        }
    }
    for (int k = 0; k < 5; ++k) w[k] = A[k][k];
}

// thread = locus: gradient and Hessian of lnL in the five log-rates from the stencil, K = -H with its spectrum floored
__global__ __launch_bounds__(64) void screen_hessian_kernel(ScreenState S) {
    const int l = blockIdx.x * blockDim.x + threadIdx.x;
    if (l >= S.L) return;
    const double* f = S.C.out + (size_t)l * kScreenPoints;
    const double h = kScreenStep, f0 = f[0];
    double H[5][5], V[5][5], w[5];
    for (int i = 0; i < 5; ++i) {
        const double fp = f[1 + 2 * i], fm = f[2 + 2 * i];
        S.g5[(size_t)l * 5 + i] = (fp - fm) / (2 * h);
        H[i][i] = -(fp - 2 * f0 + fm) / (h * h);
    }
    int qd = 0;
    for (int i = 0; i < 5; ++i)
        for (int j = i + 1; j < 5; ++j) {
            const double fpp = f[11 + 2 * qd], fmm = f[12 + 2 * qd];
            const double fi = f[1 + 2 * i] + f[2 + 2 * i], fj = f[1 + 2 * j] + f[2 + 2 * j];
            H[i][j] = H[j][i] = -(fpp + fmm - fi - fj + 2 * f0) / (2 * h * h);
            ++qd;
        }
    jacobi5(H, V, w);
    // absolute floor 1e-3: a direction along which lnL changes by less than 5e-4 per unit of log-rate is flat for every
    // purpose; on a locus without data the stencil's gradient is rounding noise, which a smaller floor amplifies
    double wmax = 0;
    for (int k = 0; k < 5; ++k) wmax = fmax(wmax, fabs(w[k]));
    const double floor_ = fmax(1e-6 * wmax, 1e-3);
    for (int i = 0; i < 5; ++i)
        for (int j = 0; j < 5; ++j) {
            double s = 0;
            for (int k = 0; k < 5; ++k) s += V[i][k] * fmax(w[k], floor_) * V[j][k];
            S.K[(size_t)l * 25 + i * 5 + j] = s;
        }
    S.best[l] = S.grm_lnl[l] - 5.0;
}

// exchangeabilities of model m at class rates theta, candidate fields for locus l
__device__ inline void sub_emit(const CandArrays& C, int64_t slot, int l, int prob, const int8_t* cls, const double* theta, const double* w6) {
    double tf = 0.0;
#pragma unroll
    for (int q = 0; q < 6; ++q) {
        const int c = cls[q];
        const double r = c < 0 ? 1.0 : exp(theta[c]);
        C.exch[slot * 6 + q] = r;
        tf = fma(r, w6[q], tf);
    }
    C.locus[slot] = l; C.vec[slot] = l; C.scale[slot] = 1.0 / tf; C.pidx[slot] = -1; C.pfac[slot] = 1.0; C.prob[slot] = prob;
}

// thread = (locus, model): constrained optimum of the quadratic model, theta = (A'KA)^-1 A'(K rho* + g); candidate at theta
__global__ __launch_bounds__(256) void screen_model_kernel(ScreenState S) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)S.L * kSubModels) return;
    const int l = (int)(i / kSubModels), m = (int)(i % kSubModels);
    const int8_t* cls = S.cls + m * 6;
    const int k = S.kk[m];
    const int free6[5] = {0, 2, 3, 4, 5};
    const double* K = S.K + (size_t)l * 25;
    double rho[5], v[5];
    for (int a = 0; a < 5; ++a) rho[a] = log(S.grm_exch[(size_t)l * 6 + free6[a]]);
    for (int a = 0; a < 5; ++a) {   // v = K rho* + g
        double s = S.g5[(size_t)l * 5 + a];
        for (int b = 0; b < 5; ++b) s = fma(K[a * 5 + b], rho[b], s);
        v[a] = s;
    }
    double M[4][5];   // augmented A'KA | A'v
    for (int a = 0; a < 4; ++a) for (int b = 0; b < 5; ++b) M[a][b] = 0.0;
    for (int a = 0; a < 5; ++a) {
        const int ca = cls[free6[a]];
        if (ca < 0) continue;
        M[ca][4] += v[a];
        for (int b = 0; b < 5; ++b) {
            const int cb = cls[free6[b]];
            if (cb >= 0) M[ca][cb] += K[a * 5 + b];
        }
    }
    double th[4] = {0, 0, 0, 0}, hdg[4] = {NAN, NAN, NAN, NAN};
    for (int a = 0; a < k; ++a) hdg[a] = M[a][a];
    // Gaussian elimination with partial pivoting on the k x k system (symmetric positive definite by construction)
    for (int c = 0; c < k; ++c) {
        int piv = c;
        for (int r = c + 1; r < k; ++r) if (fabs(M[r][c]) > fabs(M[piv][c])) piv = r;
        if (piv != c) for (int b = 0; b < 5; ++b) { const double tmp = M[c][b]; M[c][b] = M[piv][b]; M[piv][b] = tmp; }
        const double inv = 1.0 / M[c][c];
        for (int r = c + 1; r < k; ++r) {
            const double fct = M[r][c] * inv;
            for (int b = c; b < 5; ++b) M[r][b] -= fct * M[c][b];
        }
    }
    for (int c = k - 1; c >= 0; --c) {
        double s = M[c][4];
        for (int b = c + 1; b < k; ++b) s -= M[c][b] * th[b];
        th[c] = s / M[c][c];
    }
    for (int a = 0; a < 4; ++a) {
        th[a] = a < k ? fmax(fmin(th[a], kLogRateMax), kLogRateMin) : 0.0;
        S.theta[i * 4 + a] = th[a];
        S.hdiag[i * 4 + a] = hdg[a];
    }
    sub_emit(S.C, i, l, (int)i, cls, th, S.w6 + (size_t)l * 6);
}

// thread = (locus, model): true likelihood at the screened optimum -> best Akaike score per locus
__global__ __launch_bounds__(256) void screen_score_kernel(ScreenState S) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)S.L * kSubModels) return;
    const int l = (int)(i / kSubModels), m = (int)(i % kSubModels);
    const double fa = -S.C.out[i];
    S.f_at[i] = fa;
    atomic_max_double(&S.best[l], -fa - (double)S.kk[m]);
}

// A model that trails the best of its locus by more than kPruneNats even after crediting three times what the quadratic
// model missed at its screened optimum is never fitted; the others are listed for the optimiser.
__global__ __launch_bounds__(256) void screen_keep_kernel(ScreenState S, uint8_t* flags, int prune) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)S.L * kSubModels) return;
    const int l = (int)(i / kSubModels), m = (int)(i % kSubModels);
    const int8_t* cls = S.cls + m * 6;
    const int free6[5] = {0, 2, 3, 4, 5};
    const double* K = S.K + (size_t)l * 25;
    double delta[5];
    for (int a = 0; a < 5; ++a) {
        const int c = cls[free6[a]];
        delta[a] = (c >= 0 ? S.theta[i * 4 + c] : 0.0) - log(S.grm_exch[(size_t)l * 6 + free6[a]]);
    }
    double pred = -S.grm_lnl[l];
    for (int a = 0; a < 5; ++a) {
        double s = 0;
        for (int b = 0; b < 5; ++b) s = fma(K[a * 5 + b], delta[b], s);
        pred += delta[a] * (0.5 * s - S.g5[(size_t)l * 5 + a]);
    }
    const double fa = S.f_at[i], score = -fa - (double)S.kk[m];
    const bool keep = !prune || score + 3.0 * fabs(fa - pred) >= S.best[l] - kPruneNats;
    flags[i] = keep ? 1 : 0;   // compacted in index order by the driver (hipcub::DeviceSelect)
}

// ---------------------------------------------------------------------------------------------------------------------
// The surviving rate-class models (bf:542-661): at most four free class rates each, branch lengths = stash / totalFactor
// (bf:613-619).  Dense BFGS on the inverse metric started from the screen's curvature, central-difference gradients of the
// value kernel (1 + 2k evaluations), Armijo backtracking; thread = problem.
// ---------------------------------------------------------------------------------------------------------------------
struct SubState {
    int32_t Q;
    const int32_t* id;        // [Q] locus * 202 + model
    const int8_t* cls;        // [202][6]
    const int32_t* kk;        // [202]
    const double* w6;         // [L][6]
    double h;                 // difference step in log-rate
    double *x, *g, *d, *xt;   // [Q][4]
    double* Hinv;             // [Q][16] inverse metric, learned by BFGS updates
    double* H0;               // [Q][16] its start: (A'KA)^-1 from the screen's curvature K of the locus
    const double* f_known;    // [L*202] -lnL at the screened optimum (the centre of the first stencil)
    double *f, *fnew, *t, *gd, *last_df, *df;   // [Q]
    int32_t *kicks, *iters, *ls_round, *slot;
    uint8_t* phase;
    double* best;             // [L]
    int32_t* counters;
    CandArrays C;
};

// Start of a fit: the screened optimum theta_m, and as inverse metric the FULL (A'KA)^-1 of the quadratic model the optimum
// came from (round 2 kept its diagonal only): the first step is a Newton step, and the model's prediction of what a step can
// still gain is trustworthy enough to stop a fit before its first step (last_df starts at 0, not infinity).
__global__ __launch_bounds__(256) void sub_init_kernel(SubState S, const double* theta, const double* K) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= S.Q) return;
    const int id = S.id[p], l = id / kSubModels, m = id % kSubModels, k = S.kk[m];
    const int8_t* cls = S.cls + m * 6;
    const int free6[5] = {0, 2, 3, 4, 5};
    const double* Kl = K + (size_t)l * 25;
    double M[4][8];   // A'KA | I, Gauss-Jordan
    for (int a = 0; a < 4; ++a) for (int b = 0; b < 8; ++b) M[a][b] = (b - 4 == a) ? 1.0 : 0.0;
    for (int a = 0; a < 5; ++a) {
        const int ca = cls[free6[a]];
        if (ca < 0) continue;
        for (int b = 0; b < 5; ++b) {
            const int cb = cls[free6[b]];
            if (cb >= 0) M[ca][cb] += Kl[a * 5 + b];
        }
    }
    for (int a = k; a < 4; ++a) M[a][a] = 1.0;   // unused dimensions: identity
    for (int c = 0; c < 4; ++c) {
        int piv = c;
        for (int r = c + 1; r < 4; ++r) if (fabs(M[r][c]) > fabs(M[piv][c])) piv = r;
        if (piv != c) for (int b = 0; b < 8; ++b) { const double tmp = M[c][b]; M[c][b] = M[piv][b]; M[piv][b] = tmp; }
        const double inv = 1.0 / M[c][c];
        for (int b = 0; b < 8; ++b) M[c][b] *= inv;
        for (int r = 0; r < 4; ++r) {
            if (r == c) continue;
            const double fct = M[r][c];
            for (int b = 0; b < 8; ++b) M[r][b] -= fct * M[c][b];
        }
    }
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        S.x[(size_t)a * S.Q + p] = theta[(size_t)id * 4 + a];
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const double v = (a < k && b < k) ? M[a][4 + b] : (a == b ? 1.0 : 0.0);
            S.Hinv[(size_t)(a * 4 + b) * S.Q + p] = v;
            S.H0[(size_t)(a * 4 + b) * S.Q + p] = v;
        }
    }
    S.last_df[p] = 0.0; S.df[p] = 0.0; S.kicks[p] = 0; S.iters[p] = 0; S.phase[p] = PH_INIT;   // INIT: gradient wanted at x, value known
}

__global__ __launch_bounds__(256) void sub_direction_kernel(SubState S) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= S.Q || S.phase[p] != PH_LIVE) return;
    const int id = S.id[p], k = S.kk[id % kSubModels];
    double g[4], d[4], x[4];
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        g[a] = a < k ? S.g[(size_t)a * S.Q + p] : 0.0;
        x[a] = S.x[(size_t)a * S.Q + p];
        // a rate AT a bound whose gradient points out of the box is fixed for this iteration (projected gradient): without
        // this a fit whose optimum is at the bound never meets the stopping test and runs to the iteration limit
        if ((x[a] <= kLogRateMin && g[a] > 0.0) || (x[a] >= kLogRateMax && g[a] < 0.0)) g[a] = 0.0;
    }
    double gd = 0, gg = 0, gmax = 0;
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        double s = 0;
#pragma unroll
        for (int b = 0; b < 4; ++b) s = fma(S.Hinv[(size_t)(a * 4 + b) * S.Q + p], g[b], s);
        d[a] = a < k ? fmax(fmin(-s, kMaxLogStep), -kMaxLogStep) : 0.0;
        gd = fma(g[a], d[a], gd); gg = fma(g[a], g[a], gg); gmax = fmax(gmax, fabs(g[a]));
    }
    if (!(gd < 0)) {
#pragma unroll
        for (int a = 0; a < 4; ++a) d[a] = -g[a];
        gd = -gg;
    }
    const double f = S.f[p], scale_f = 1.0 + fabs(f), last_df = S.last_df[p];
    // Boundary trap of the log scale, tested on EVERY iteration: a class rate near its lower bound (typically inherited from a
    // general model that put that rate at zero) although the likelihood rises with the rate itself.  In log r both the gradient
    // r f' and the curvature vanish there, a quasi-Newton search creeps (seen on a 3-taxon locus: 100 iterations to move a
    // rate from 9e-4 to 2e-3 when its optimum was 0.35, 0.03-0.1 lnL short on 21 models); the rate is put at kEscapeRate and
    // the search restarts from there with the start metric -- at most twice per fit (a rate that comes back really is zero).
    {
        bool moved = false;
        const bool few = S.kicks[p] < 2;
#pragma unroll
        for (int a = 0; a < 4; ++a)
            if (a < k && x[a] < -3.7 /* r < kEscapeRate / 2 */ && (g[a] / exp(x[a])) * kEscapeRate < -1e-7 * scale_f && few) {
                S.x[(size_t)a * S.Q + p] = log(kEscapeRate);
                moved = true;
            }
        if (moved) {
            S.kicks[p] += 1;
#pragma unroll
            for (int a = 0; a < 16; ++a) S.Hinv[(size_t)a * S.Q + p] = S.H0[(size_t)a * S.Q + p];
            S.last_df[p] = INFINITY;
            S.phase[p] = PH_RESTART;
            atomicAdd(&S.counters[C_NGRAD], 1);
            return;
        }
    }
    // Stop when the quasi-Newton model predicts that a further step gains nothing.  Unlike the general model's rule there is
    // no condition on what the LAST step gained: the metric here starts as the Hessian of the screen's quadratic model, so
    // its prediction is trustworthy as soon as it exists -- a fit that one Newton step has brought to its optimum is not
    // made to take a second one to prove it (last_df = infinity, after a reset of the metric, still forces a step).
    // (a tenth of the general model's tolerance: the screen floors the curvature of nearly flat directions at 1e-3, where the
    // model then under-predicts what a step gains -- seen: 5e-5 gained against 2e-6 predicted)
    const bool stop = (last_df < INFINITY && -gd <= 0.1 * kPtol * scale_f && gmax <= kGtol * scale_f) || gmax <= 1e-9;
    if (stop) {
        S.phase[p] = PH_CONV;
        atomicSub(&S.counters[C_NLIVE], 1);
        return;
    }
    double dmax = 0;
#pragma unroll
    for (int a = 0; a < 4; ++a) { S.d[(size_t)a * S.Q + p] = d[a]; dmax = fmax(dmax, fabs(d[a])); }
    S.t[p] = fmin(1.0, kMaxLogStep / fmax(dmax, 1e-300));
    S.gd[p] = gd;
    S.ls_round[p] = 0;
    S.phase[p] = PH_PEND;
    atomicAdd(&S.counters[C_NPEND], 1);
}

__global__ __launch_bounds__(256) void sub_trial_kernel(SubState S) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= S.Q || S.phase[p] != PH_PEND) return;
    const int id = S.id[p], l = id / kSubModels, m = id % kSubModels;
    const double t = S.t[p];
    double xt[4];
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        xt[a] = fmax(fmin(fma(t, S.d[(size_t)a * S.Q + p], S.x[(size_t)a * S.Q + p]), kLogRateMax), kLogRateMin);
        S.xt[(size_t)a * S.Q + p] = xt[a];
    }
    const int slot = atomicAdd(&S.counters[C_NCAND], 1);
    S.slot[p] = slot;
    sub_emit(S.C, slot, l, p, S.cls + m * 6, xt, S.w6 + (size_t)l * 6);
}

__global__ __launch_bounds__(256) void sub_accept_kernel(SubState S, int ncand) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= ncand) return;
    const int p = S.C.prob[c];
    const double ft = -S.C.out[c], f = S.f[p], t = S.t[p], gd = S.gd[p];
    if (isfinite(ft) && ft <= f + 1e-4 * t * gd) {
        S.fnew[p] = ft;
        S.phase[p] = PH_ACCEPTED;
        atomicAdd(&S.counters[C_NGRAD], 1);
    } else if (++S.ls_round[p] >= kMaxLineSearch) {
        S.phase[p] = PH_FAILED;
        atomicAdd(&S.counters[C_NFAILED], 1);
    } else {
        S.t[p] = backtrack(t, ft, f, gd);
        atomicAdd(&S.counters[C_NPEND], 1);
    }
}

// difference stencil around the accepted (or moved, or initial) point: 1 + 2k candidates
__global__ __launch_bounds__(256) void sub_stencil_kernel(SubState S) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= S.Q) return;
    const uint8_t ph = S.phase[p];
    if (ph != PH_ACCEPTED && ph != PH_RESTART && ph != PH_INIT) return;
    const int id = S.id[p], l = id / kSubModels, m = id % kSubModels, k = S.kk[m];
    const double* X = ph == PH_ACCEPTED ? S.xt : S.x;
    double x[4];
#pragma unroll
    for (int a = 0; a < 4; ++a) x[a] = X[(size_t)a * S.Q + p];
    // slot = where the centre point sits (or would sit: at INIT its value is known from the screen and not re-evaluated)
    const int base = ph == PH_INIT ? atomicAdd(&S.counters[C_NCAND], 2 * k) - 1 : atomicAdd(&S.counters[C_NCAND], 1 + 2 * k);
    S.slot[p] = base;
    const int8_t* cls = S.cls + m * 6;
    const double* w6 = S.w6 + (size_t)l * 6;
    if (ph != PH_INIT) sub_emit(S.C, base, l, p, cls, x, w6);
    for (int a = 0; a < k; ++a) {
        const double keep = x[a];
        x[a] = keep + S.h; sub_emit(S.C, base + 1 + 2 * a, l, p, cls, x, w6);
        x[a] = keep - S.h; sub_emit(S.C, base + 2 + 2 * a, l, p, cls, x, w6);
        x[a] = keep;
    }
}

__global__ __launch_bounds__(256) void sub_update_kernel(SubState S, int prune) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= S.Q) return;
    const uint8_t ph = S.phase[p];
    const int id = S.id[p], l = id / kSubModels, m = id % kSubModels, k = S.kk[m];
    double H[16];
#pragma unroll
    for (int a = 0; a < 16; ++a) H[a] = S.Hinv[(size_t)a * S.Q + p];
    if (ph == PH_FAILED) {
        // a failed line search with a learned metric: fall back to the screen's metric once before giving up
        bool learned = false;
#pragma unroll
        for (int a = 0; a < 16; ++a) learned |= H[a] != S.H0[(size_t)a * S.Q + p];
        S.iters[p] += 1;
        S.df[p] = 0.0;
        if (learned) {
#pragma unroll
            for (int a = 0; a < 16; ++a) S.Hinv[(size_t)a * S.Q + p] = S.H0[(size_t)a * S.Q + p];
            S.last_df[p] = INFINITY;
            S.phase[p] = PH_LIVE;
        } else {
            S.last_df[p] = 0.0;
            S.phase[p] = PH_CONV;
            atomicSub(&S.counters[C_NLIVE], 1);
        }
        return;
    }
    if (ph != PH_ACCEPTED && ph != PH_RESTART && ph != PH_INIT) return;
    const int base = S.slot[p];
    const double fx = ph == PH_INIT ? S.f_known[id] : -S.C.out[base];
    double gx[4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
        gx[a] = a < k ? (-S.C.out[base + 1 + 2 * a] + S.C.out[base + 2 + 2 * a]) / (2.0 * S.h) : 0.0;
    if (ph == PH_RESTART || ph == PH_INIT) {
#pragma unroll
        for (int a = 0; a < 4; ++a) S.g[(size_t)a * S.Q + p] = gx[a];
        S.f[p] = fx;
        S.df[p] = INFINITY;   // no abandoning on a point that was not reached by a step (sub_prune_kernel)
        S.phase[p] = PH_LIVE;
        if (prune) atomic_max_double(&S.best[l], -fx - (double)k);
        return;
    }
    double s[4], y[4], Hy[4];
    double sy = 0, ss = 0, yy = 0;
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        s[a] = S.xt[(size_t)a * S.Q + p] - S.x[(size_t)a * S.Q + p];
        y[a] = gx[a] - S.g[(size_t)a * S.Q + p];
        sy = fma(s[a], y[a], sy); ss = fma(s[a], s[a], ss); yy = fma(y[a], y[a], yy);
    }
    if (sy > 1e-12 * sqrt(ss * yy + 1e-300)) {   // dense BFGS update of the inverse metric
        const double rho = 1.0 / sy;
        double yHy = 0;
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            double t2 = 0;
#pragma unroll
            for (int b = 0; b < 4; ++b) t2 = fma(H[a * 4 + b], y[b], t2);
            Hy[a] = t2;
        }
#pragma unroll
        for (int a = 0; a < 4; ++a) yHy = fma(y[a], Hy[a], yHy);
        const double c1 = (1.0 + rho * yHy) * rho;
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b)
                S.Hinv[(size_t)(a * 4 + b) * S.Q + p] = H[a * 4 + b] + c1 * s[a] * s[b] - rho * (Hy[a] * s[b] + s[a] * Hy[b]);
    }
    const double df = S.f[p] - fx;
    S.last_df[p] = df;
    S.df[p] = df;
#pragma unroll
    for (int a = 0; a < 4; ++a) { S.x[(size_t)a * S.Q + p] = S.xt[(size_t)a * S.Q + p]; S.g[(size_t)a * S.Q + p] = gx[a]; }
    S.f[p] = fx;
    S.iters[p] += 1;
    S.phase[p] = PH_LIVE;
    if (prune) atomic_max_double(&S.best[l], -fx - (double)k);
}

// Models that cannot matter are abandoned: Akaike score trailing the best of the locus by more than kPruneNats even after
// crediting three times the last improvement (weight below e^-21: invisible in the averaged rates)
__global__ __launch_bounds__(256) void sub_prune_kernel(SubState S) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= S.Q || S.phase[p] != PH_LIVE) return;
    const int id = S.id[p], l = id / kSubModels, k = S.kk[id % kSubModels];
    const double score = -S.f[p] - (double)k;
    if (score + 3.0 * fmax(S.df[p], 0.0) < S.best[l] - kPruneNats) {
        S.phase[p] = PH_CONV;
        atomicSub(&S.counters[C_NLIVE], 1);
        atomicAdd(&S.counters[C_NPRUNED], 1);
    }
}

// fitted problems back into the per-(locus, model) tables
__global__ __launch_bounds__(256) void sub_scatter_kernel(SubState S, double* theta, double* f_at, int32_t* iters_out) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= S.Q) return;
    const int id = S.id[p];
#pragma unroll
    for (int a = 0; a < 4; ++a) theta[(size_t)id * 4 + a] = S.x[(size_t)a * S.Q + p];
    f_at[id] = S.f[p];
    if (iters_out) iters_out[id] = S.iters[p];
}

// ---------------------------------------------------------------------------------------------------------------------
// Akaike weights and model-averaged rates (bf:806-847): w_m proportional to exp(lnL_m - k_m); thread = locus
// ---------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void average_kernel(int32_t L, const double* grm_exch, const double* grm_lnl, const double* theta,
                                                     const double* f_at, const int8_t* cls, const int32_t* kk, double* exch_out,
                                                     double* weights, double* lnl, double* model_exch) {
    const int l = blockIdx.x * blockDim.x + threadIdx.x;
    if (l >= L) return;
    double smax = grm_lnl[l] - 5.0;
    for (int m = 0; m < kSubModels; ++m) smax = fmax(smax, -f_at[(size_t)l * kSubModels + m] - (double)kk[m]);
    double wsum = 0.0, avg[6] = {0, 0, 0, 0, 0, 0};
    for (int m = 0; m < kModels; ++m) {
        double ll, r[6];
        int k;
        if (m == 0) {
            ll = grm_lnl[l]; k = 5;
            for (int q = 0; q < 6; ++q) r[q] = grm_exch[(size_t)l * 6 + q];
        } else {
            const size_t i = (size_t)l * kSubModels + (m - 1);
            ll = -f_at[i]; k = kk[m - 1];
            for (int q = 0; q < 6; ++q) { const int c = cls[(m - 1) * 6 + q]; r[q] = c < 0 ? 1.0 : exp(theta[i * 4 + c]); }
        }
        const double w = exp(ll - (double)k - smax);
        wsum += w;
        for (int q = 0; q < 6; ++q) avg[q] = fma(w, r[q], avg[q]);
        if (weights) weights[(size_t)l * kModels + m] = w;
        if (lnl) lnl[(size_t)l * kModels + m] = ll;
        if (model_exch) for (int q = 0; q < 6; ++q) model_exch[((size_t)l * kModels + m) * 6 + q] = r[q];
    }
    for (int q = 0; q < 6; ++q) exch_out[(size_t)l * 6 + q] = q == 1 ? 1.0 : avg[q] / wsum;
    if (weights) for (int m = 0; m < kModels; ++m) weights[(size_t)l * kModels + m] /= wsum;
}

}  // namespace s1
}  // namespace tphip
