/*
 * tphip.h -- C ABI of libtphip.so: the MI355X (gfx950) site-rate + phylogenetic-informativeness engine.
 *
 * This is the drop-in boundary for ONE hot path of faircloth-lab/tapir: what `worker()` in
 * bin/tapir_compute.py:84-123 does per locus -- the shell-out to HyPhy (per-site substitution-rate ML,
 * tapir/data/models_and_rates.bf:978-1070) followed by Townsend's PI(t), its per-locus sums and its
 * scipy.integrate.quad interval integrals (tapir/compute.py:46-52, 76-94, 96-110) -- computed for ALL
 * loci in one column-parallel batch on the GPU.
 *
 * Conventions
 *   - extern "C", plain pointers and sizes only.  No exceptions cross the boundary.
 *   - every function returns 0 on success, non-zero on error; tphip_last_error() gives the message
 *     (thread-local, owned by the library).
 *   - there is NO CPU backend: a call fails with TPHIP_ERR_NO_DEVICE when no gfx950 device is usable.
 *   - the caller owns every data buffer.  `*_dev` entry points take DEVICE pointers (the hot path: inputs
 *     already resident in HBM, nothing leaves the device); the plain entry points take HOST pointers and
 *     do the H2D/D2H copies themselves.  The library keeps no pointer past the return of a host-pointer
 *     call; a plan keeps only its own small device tables (tree program, per-locus models, schedule).
 *   - state codes: one byte per alignment cell, bit mask A=1 C=2 G=4 T=8; gap/?/N = 15; IUPAC codes are
 *     unions (HyPhy resolves ambiguities as sets in the likelihood and fractionally in
 *     HarvestFrequencies, bf:968).  Layout is TAXON-MAJOR over the flattened batch:
 *     states[taxon * ncols_total + column]; loci are concatenated along the column axis and
 *     locus_offsets[L+1] gives their column ranges.  A wave reads 64 consecutive bytes of one taxon row.
 *   - tree: nodes in post-order (children before parents, root last); parent[root] = -1; leaf_taxon[n] =
 *     alignment row of leaf n, -1 for internal nodes; branch_len[n] = length of the branch above n,
 *     ALREADY divided by the correction factor (tapir/compute.py:59-74).  Multifurcations allowed.
 *   - PI table row for a locus: [ net PI at t=0..T-1 | PI at each of n_t times | sum(integral) per
 *     interval | sum(error) per interval ],  W = T + n_t + 2*n_i doubles (what tapir/db.py:44-61 stores).
 */
#ifndef TPHIP_H
#define TPHIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TPHIP_VERSION 110 /* 0.1.1: tphip_plan_desc.struct_size, TPHIP_START_AUTO, tphip_stage1_fit, tphip_plan_set_models */

enum {
    TPHIP_OK = 0,
    TPHIP_ERR_INVALID = 1,   /* bad argument (message says which) */
    TPHIP_ERR_NO_DEVICE = 2, /* no usable GPU: the library has no CPU path */
    TPHIP_ERR_HIP = 3,       /* a HIP runtime call failed */
    TPHIP_ERR_WORKSPACE = 4  /* caller's workspace is too small */
};

/* per-column flags written by the site-rate kernel */
enum {
    TPHIP_FLAG_OK = 0,        /* interior optimum                                                   */
    TPHIP_FLAG_FLAT = 1,      /* <= 1 resolved taxon: L independent of the rate; s stays 1 (bf:1050)  */
    TPHIP_FLAG_SATURATED = 2, /* log L flat to fp64 on the way to s -> inf (or uphill at s = 1e4): s = 1e4 */
    TPHIP_FLAG_ZERO = 3,      /* all resolved taxa share one base: optimum exactly at s = 0          */
    TPHIP_FLAG_MAXIT = 4      /* iteration limit                                                     */
};

/* integral modes for the interval sums (tapir/compute.py:50-52, 81-94) */
enum {
    TPHIP_INTEG_QUADPACK = 0, /* emulates scipy.integrate.quad: QUADPACK dqagse, GK21, eps 1.49e-8, limit 50;
                                 also yields sum(error) for the sqlite `interval.error` column          */
    TPHIP_INTEG_CLOSED = 1    /* analytic antiderivative -(4rt+1)exp(-4rt); error column = 0            */
};

/* tphip_plan_desc.start_rule: where the per-site optimiser starts (HyPhy: siteRate = 1 before every Optimize, bf:1050) */
enum {
    TPHIP_START_AUTO = 0,      /* default: HyPhy's start on trees of fewer than 32 taxa, the parsimony start from 32 on
                                  (and with the rate-mixture extension, which HyPhy's script does not have)                */
    TPHIP_START_REFERENCE = 1, /* siteRate = 1 on every tree                                                            */
    TPHIP_START_PARSIMONY = 2  /* the column's parsimony rate on every tree (one evaluation fewer per column)           */
};

/* tphip_plan_desc.pattern_dedup */
enum { TPHIP_DEDUP_AUTO = 0, TPHIP_DEDUP_OFF = 1, TPHIP_DEDUP_ON = 2 };

int tphip_version(void);
const char *tphip_last_error(void);
/* number of usable HIP devices (0 if none; never fails) */
int tphip_device_count(void);

/* ------------------------------------------------------------------------------------------------
 * Plan: everything that is small and shared by a batch -- the tree's traversal program, the per-locus
 * GTR models (eigen-systems are computed on the device), the column ranges and the PI schedule.
 * Replaces: the three stdin lines + in-script constants HyPhy is configured with (bf:4-9, 966-1013) and
 * the per-locus `params` list of bin/tapir_compute.py:148-152.
 * ---------------------------------------------------------------------------------------------- */
typedef struct tphip_plan tphip_plan;

typedef struct tphip_plan_desc {
    uint32_t struct_size;      /* sizeof(tphip_plan_desc) as the caller compiled it.  Zero-initialise the struct, then set
                                  this: fields the caller's header did not have yet take their zero defaults instead of being
                                  read from beyond the caller's struct                                   */
    int32_t device;            /* HIP device ordinal (>= 0)                                            */
    /* tree (host pointers) */
    int32_t ntaxa;             /* rows of the alignment                                                */
    int32_t nnodes;            /* nodes of the tree, post-order                                        */
    const int32_t *parent;     /* [nnodes]                                                             */
    const double *branch_len;  /* [nnodes], already / correction                                       */
    const int32_t *leaf_taxon; /* [nnodes]                                                             */
    /* loci (host pointers) */
    int64_t nloci;
    const int64_t *locus_offsets; /* [nloci+1] column ranges in the flattened batch                    */
    const double *pi;          /* [nloci*4] A,C,G,T base frequencies (bf:968 HarvestFrequencies)       */
    const double *exch;        /* [nloci*6] AC,AG,AT,CG,CT,GT exchangeabilities (bf:970-976; AG == 1)  */
    /* PI schedule */
    int32_t T;                 /* net PI is evaluated at t = 0..T-1, T = int(tree depth) (compute.py:54-57) */
    const int32_t *times;      /* [n_t] --times, each < T (compute.py:76-79)                           */
    int32_t n_t;
    const int32_t *intervals;  /* [n_i*2] --intervals as (start, stop), start < stop (compute.py:90-91) */
    int32_t n_i;
    int32_t integ_mode;        /* TPHIP_INTEG_*                                                        */
    /* rate post-processing between the two stages (bin/tapir_compute.py:100-102) */
    double correction;         /* rates are divided by this (parse_site_rates, compute.py:38-39)       */
    int32_t threshold;         /* columns with fewer A/C/G/T cells become NaN (compute.py:96-110)      */
    int32_t round_decimals;    /* 4 = round rates as HyPhy's Format(x,0,4) does before PI (bf:1093-1095);
                                  < 0 = keep full precision                                            */
    /* Opt-in extension, NOT in the reference (its script has no rate mixture): a discrete mixture of
     * rate categories on top of the per-site rate, L(s) = sum_k cat_weight[k] L(s * cat_rate[k]) -- the "+G" of
     * GTR+G when the categories are the discrete gamma of Yang (1994).  ncat <= 1 = the reference's model. */
    int32_t ncat;              /* number of categories, at most 16                                    */
    const double *cat_rate;    /* [ncat] rate multipliers > 0 (mean 1 keeps `rate` = kappa * s interpretable) */
    const double *cat_weight;  /* [ncat] weights > 0 (normalised by the library)                       */
    /* Where the per-site optimiser starts.  HyPhy starts every column at siteRate = 1 (bf:1050) and returns the local
     * optimum uphill of that point.  The parsimony start (the column's parsimony rate) saves one evaluation
     * per column and reaches the same maximum on every unimodal column; on the rare multimodal ones it may end on another
     * local optimum than a search from 1 would -- measured: 0 of 4e5 columns at 64 taxa, 6e-5 at 16 taxa, 1e-3 of noisy
     * 5-taxon columns.  TPHIP_START_AUTO therefore uses it only on trees of 32 taxa or more, where no
     * deviation was ever seen, and HyPhy's own start below; the other two values force one rule on every tree. */
    int32_t start_rule;        /* TPHIP_START_*                                                        */
    /* HyPhy fits one rate per UNIQUE column pattern of a locus and reports it for every column that carries it
     * (bf:1033-1044: GetDataInfo(dupInfo...), alreadyDone[siteMap]).  TPHIP_DEDUP_AUTO does the same wherever a cheap
     * per-locus estimate says at least ~15 % of the columns that need the optimiser are repeats (real loci; the synthetic
     * alignments with random gaps hardly repeat a column), in batches of at least 2^20 columns (smaller ones are
     * latency-bound: nothing to gain); ON / OFF force it.  The outputs are bit-identical either way. */
    int32_t pattern_dedup;     /* TPHIP_DEDUP_*                                                        */
} tphip_plan_desc;

int tphip_plan_create(const tphip_plan_desc *desc, tphip_plan **out);
int tphip_plan_destroy(tphip_plan *plan);
/* width W of a PI table row, and total columns */
int32_t tphip_plan_table_width(const tphip_plan *plan);
int64_t tphip_plan_ncols(const tphip_plan *plan);
/* bytes of device scratch the *_dev entry points need (compacted column list, per-chunk PI partials) */
size_t tphip_plan_workspace_bytes(const tphip_plan *plan);
/* chronogram length = sum of branch lengths (bf:1006-1013) and LDS stack depth of the program */
double tphip_plan_chrono_length(const tphip_plan *plan);
int32_t tphip_plan_stack_depth(const tphip_plan *plan);
/* op mix of the compiled tree program: counts[5] = TIP_SET, TIP_MUL, BRANCH, PUSH, POP_MUL (for FLOP models) */
int tphip_plan_op_counts(const tphip_plan *plan, int32_t *counts);
/* number of TIP_SET + TIP_MUL pairs the site-rate kernel executes as one fused CHERRY op (equal branch lengths:
 * one set of exponentials for both tips) */
int32_t tphip_plan_cherry_count(const tphip_plan *plan);
/* copy the per-locus eigen-systems back (tests): lam[L*4], U[L*16], Uinv[L*16], kappa[L] */
int tphip_plan_get_models(const tphip_plan *plan, double *lam, double *U, double *Uinv, double *kappa);
/* Replace the per-locus models of an existing plan: pi [nloci*4] and / or exch [nloci*6] (host; NULL keeps the current
 * values).  The eigen-systems are recomputed on the device.  What it is for: one plan per batch through both stages of
 * HyPhy's script -- tphip_stage1_fit estimates the exchangeabilities (bf:405-897), the per-site loop then runs on them
 * (bf:970-976) -- instead of a second plan.  Same validation as tphip_plan_create. */
int tphip_plan_set_models(tphip_plan *plan, const double *pi, const double *exch);

/* ------------------------------------------------------------------------------------------------
 * Device-pointer entry points (the hot path).  `stream` is a hipStream_t (NULL = default stream).
 * Kernels are enqueued and the call returns without synchronising.
 * ---------------------------------------------------------------------------------------------- */

/* Stage 1 -- replaces Popen([hyphy, template]) for every locus at once (bin/tapir_compute.py:92-99).
 * Per column: rate = kappa*s_hat, subst = rate*chronoLength, lnl, flag, nres = #A/C/G/T cells. */
int tphip_site_rates_dev(tphip_plan *plan, const uint8_t *d_states, double *d_rate, double *d_subst,
                         double *d_lnl, uint8_t *d_flag, int32_t *d_nres, void *d_workspace,
                         size_t workspace_bytes, void *stream);

/* Stage 2 -- replaces get_townsend_pi + nansum + get_net_pi_for_periods + get_net_integral_for_epochs
 * (bin/tapir_compute.py:114-122).  d_rates are the raw stage-1 rates; rounding, /correction and the
 * informative-site cull (d_nres may be NULL = no cull, the --site-rates path, bin/tapir_compute.py:103-104)
 * are applied on the fly.  d_tables is [nloci][W]. */
int tphip_pi_tables_dev(tphip_plan *plan, const double *d_rates, const int32_t *d_nres, double *d_tables,
                        void *d_workspace, size_t workspace_bytes, void *stream);

/* Both stages back to back on one stream: the whole worker() body for the whole batch. */
int tphip_run_dev(tphip_plan *plan, const uint8_t *d_states, double *d_rate, double *d_subst, double *d_lnl,
                  uint8_t *d_flag, int32_t *d_nres, double *d_tables, void *d_workspace,
                  size_t workspace_bytes, void *stream);

/* parse_site_rates + cull_uninformative_rates as the PI stage applies them to a raw rate (tapir/compute.py:24-44,
 * 108-110; bin/tapir_compute.py:100-102): d_out[c] = R4(d_rates[c]) / correction, NaN where d_nres[c] < threshold
 * (d_nres may be NULL: no cull -- what parse_site_rates alone returns and writes as `corrected_rates`).  R4 is the
 * round trip through HyPhy's Format(x,0,4) (bf:1093-1095): the double nearest to the decimal that printf-style
 * rounding of the exact binary value gives (skipped when round_decimals < 0). */
int tphip_corrected_rates_dev(tphip_plan *plan, const double *d_rates, const int32_t *d_nres, double *d_out,
                              void *stream);

/* tapir/compute.py:46-48 get_townsend_pi(time, rates) as a dense (n_times, n) matrix, row-major:
 * out[k*n + i] = 16 r_i^2 t_k exp(-4 r_i t_k).  NaN rates propagate (numpy semantics). */
int tphip_townsend_pi_dense_dev(int32_t device, const double *d_rates, int64_t n, const double *d_times,
                                int32_t n_times, double *d_out, void *stream);

/* tapir/compute.py:50-52 vectorised over sites: (integral, abserr) of scipy.integrate.quad(
 * get_townsend_pi, a, b, args=(rate)) for every rate (QUADPACK dqagse emulation, or the closed form with
 * abserr = 0 when integ_mode == TPHIP_INTEG_CLOSED).  Non-finite rates give NaN. */
int tphip_quad_townsend_dev(int32_t device, const double *d_rates, int64_t n, double a, double b,
                            int32_t integ_mode, double *d_integral, double *d_abserr, void *stream);

/* HarvestFrequencies(Freqs, filter, 1, 1, 1) (bf:968): per-locus counts of each of the 16 state masks;
 * d_hist is [nloci][16] int64 (pi follows on the host: each mask adds 1/popcount to its bases).
 * d_locus_offsets is a DEVICE array [nloci+1].  HBM-bound byte kernel: ntaxa bytes read per column. */
int tphip_state_histogram_dev(int32_t device, const uint8_t *d_states, int64_t ncols_total, int32_t ntaxa,
                              const int64_t *d_locus_offsets, int64_t nloci, int64_t *d_hist, void *stream);

/* Whole-locus log-likelihood for a batch of candidate parameter sets: the objective HyPhy's stage 1 maximises in
 * every `Optimize(lf_MLES, lf)` (models_and_rates.bf:487-520, 647-655): sum over the columns of locus
 * cand_locus[c] of log L(column | exchangeabilities cand_exch[c][6] = AC,AG,AT,CG,CT,GT, branch lengths, base
 * frequencies = the plan's pi for that locus), every site at rate 1.  Branch lengths (above each node, post-order
 * numbering of the plan's tree, root entry ignored) of candidate c are
 *     blen_vecs[cand_vec[c]][b] * cand_scale[c] * (b == cand_pidx[c] ? cand_pfac[c] : 1)
 * so a finite-difference stencil, or the rate-class models that only rescale stashed lengths (bf:613-619), share
 * one stored vector.  One workgroup per candidate; the optimiser runs on the host (tapir_amd/stage1.py). */
int tphip_locus_loglik_dev(tphip_plan *plan, const uint8_t *d_states, int64_t ncand, const int32_t *d_cand_locus,
                           const double *d_cand_exch, const double *d_blen_vecs, const int32_t *d_cand_vec,
                           const double *d_cand_scale, const int32_t *d_cand_pidx, const double *d_cand_pfac,
                           double *d_out, void *stream);

/* Value AND gradient of the same objective from one forward + one reverse sweep of the pruning recursion
 * (reverse-mode differentiation; replaces a finite-difference stencil of 2 x (5 + 2N-3) likelihood evaluations).
 * Candidates are described exactly as for tphip_locus_loglik_dev.  Outputs per candidate:
 *   d_lnl[c]; d_dexch[c][6] = d lnL / d (AC, AG, AT, CG, CT, GT) with branch lengths held fixed;
 *   d_dlogt[c][nnodes] = d lnL / d log t_b (root entry 0; may be NULL); d_sum_dlogt[c] = its sum over branches
 *   (the only branch-length derivative a rate-class model needs: bf:613-619 ties all lengths to one factor);
 *   d_d2logt[c][nnodes] = d2 lnL / d (log t_b)^2, the diagonal of the Hessian in the branch lengths (may be NULL;
 *   the optimiser's preconditioner). */
int tphip_locus_gradient_dev(tphip_plan *plan, const uint8_t *d_states, int64_t ncand, const int32_t *d_cand_locus,
                             const double *d_cand_exch, const double *d_blen_vecs, const int32_t *d_cand_vec,
                             const double *d_cand_scale, const int32_t *d_cand_pidx, const double *d_cand_pfac,
                             double *d_lnl, double *d_dexch, double *d_dlogt, double *d_sum_dlogt, double *d_d2logt,
                             void *stream);

/* ------------------------------------------------------------------------------------------------
 * HyPhy's stage 1 as one call: model-averaged exchangeabilities for every locus of the plan.
 * Replaces, per locus, models_and_rates.bf:487-520 (general reversible model: Optimize over AC, AT, CG, CT, GT
 * and every branch length), :522-540 (branch lengths stashed as expected substitutions), :542-661 (the other 202
 * partitions of the six rates into classes, Optimize over at most four class rates each) and :806-847 (Akaike weights
 * w_m ~ exp(lnL_m - k_m), model-averaged rates).  The optimisers run on the device (stage1_opt_kernels.hpp: L-BFGS for
 * the general model, dense BFGS for the rate-class models, a quadratic screen that decides which of the 202 can carry
 * weight); the host sequences launches.  The plan's own exchangeabilities are ignored, its pi and tree are used; with
 * column weights set (tphip_plan_set_column_weights) the plan's columns are site patterns, as HyPhy evaluates them.
 * Models are indexed 0..202: 0 = "012345" (the general model), then the other restricted growth strings over
 * (AC, AG, AT, CG, CT, GT) in lexicographic order = the order of the loops at bf:544-566.
 * Outputs are HOST arrays (the call synchronises `stream` before it returns); all but exch may be NULL:
 *   exch[L][6]           model-averaged AC, AG (= 1), AT, CG, CT, GT -- what stage 2 takes (bf:838-847)
 *   pi[L][4]             the base frequencies used (the plan's, or the empirical ones: opts.empirical_pi)
 *   weights[L][203]      Akaike weights;  lnl[L][203] maximised log-likelihoods;  model_exch[L][203][6] fitted rates
 *   grm_blen[L][nnodes]  branch lengths t_b of the general model (stash / totalFactor)
 *   grm_iters[L], sub_iters[L][202]  optimiser iterations;  stats[8] = likelihood evaluations, gradients, evaluations and
 *                        gradients of the general model, models abandoned, models fitted, outer iterations (general, class) */
typedef struct tphip_stage1_opts {
    uint32_t struct_size;   /* sizeof(tphip_stage1_opts) as the caller compiled it; fields beyond it take their defaults.
                               Zero-initialise the struct, then set this.                                          */
    int32_t maxit_grm;      /* iteration limit of the general model, 0 = max(300, 4 * (5 + branches))              */
    int32_t maxit_sub;      /* iteration limit of a rate-class model, 0 = 100                                      */
    int32_t no_prune;       /* 1 = fit all 202 models to convergence (default 0: a model whose Akaike weight cannot
                               exceed e^-21 is abandoned with the likelihood it has reached, a lower bound)          */
    double fd_step;         /* step of the central differences of the rate-class fits in log-rate, 0 = 1e-4         */
    int32_t free_root_pair; /* 1 = the two branches below a bifurcating root are separate coordinates, as in HyPhy's
                               parameter list (default 0: one coordinate for their sum, which is all a reversible
                               model's likelihood depends on; the reported pair keeps the input tree's proportion)  */
    int32_t compress_patterns; /* 1 = the plan's columns are raw alignment columns: collapse them on the device into unique
                               site patterns with counts first, which is what HyPhy's likelihood function sums over
                               (GetDataInfo(dupInfo...), bf:960-963).  0 = evaluate the plan's columns as they are (with
                               the plan's column weights, if set)                                                    */
    int32_t empirical_pi;   /* 1 = base frequencies from the alignment itself, HarvestFrequencies(.., 1, 1, 1) (bf:968):
                               every cell adds 1 / popcount(mask) to each base it may be.  They replace the plan's pi
                               (as tphip_plan_set_models would) and are returned in `pi`.  0 = the plan's pi          */
    int64_t row_pitch;      /* host-pointer call only: bytes between taxon rows of `states` (0 = the plan's column count).
                               Lets a caller pass a column range of a bigger taxon-major array without copying it.   */
} tphip_stage1_opts;

int tphip_stage1_fit_dev(tphip_plan *plan, const uint8_t *d_states, const tphip_stage1_opts *opts, double *exch, double *pi,
                         double *weights, double *lnl, double *model_exch, double *grm_blen, int32_t *grm_iters,
                         int32_t *sub_iters, int64_t *stats, void *stream);
/* host-pointer twin: `states` is uploaded once (a 2-D copy when opts.row_pitch is set); d_states_cache as for
 * tphip_locus_loglik (it must be NULL when row_pitch or compress_patterns is used: nothing of the upload is kept) */
int tphip_stage1_fit(tphip_plan *plan, const uint8_t *states, void **d_states_cache, const tphip_stage1_opts *opts,
                     double *exch, double *pi, double *weights, double *lnl, double *model_exch, double *grm_blen,
                     int32_t *grm_iters, int32_t *sub_iters, int64_t *stats);

/* Profiling hooks for bench.py: when enabled the library brackets its dominant kernel (site rates) with
 * HIP events on the caller's stream and accumulates the elapsed time. */
int tphip_profile_enable(tphip_plan *plan, int32_t on);
/* sums since the last reset; *launches = number of bracketed launches; the call synchronises the events */
int tphip_profile_read(tphip_plan *plan, double *site_rate_ms, double *pi_ms, int64_t *launches, int32_t reset);
/* total likelihood evaluations (Newton iterations summed over columns) of the last site-rate launch */
int tphip_last_eval_count(tphip_plan *plan, int64_t *evals);

/* ------------------------------------------------------------------------------------------------
 * Host-pointer entry points: same stages, library does the copies (PCIe time included by construction).
 * ---------------------------------------------------------------------------------------------- */
/* Pinned host memory for the buffers of the host-pointer calls: with it the copies are direct DMA and the per-column
 * results travel while the PI kernels run; ordinary (pageable) memory works too, through the runtime's staged copies.
 * When EVERY buffer of a tphip_site_rates / tphip_run_fused call is pinned (the input `states` too) and the batch has
 * at least 2^21 columns, the library runs it as a pipeline of two locus groups (the second upload under the first
 * group's kernels); outputs are identical to the unsplit run.  NULL (and tphip_last_error) when the allocation fails.
 * Free with tphip_host_free. */
void *tphip_host_alloc(size_t bytes);
int tphip_host_free(void *ptr);
int tphip_site_rates(tphip_plan *plan, const uint8_t *states, double *rate, double *subst, double *lnl,
                     uint8_t *flag, int32_t *nres);
int tphip_pi_tables(tphip_plan *plan, const double *rates, const int32_t *nres, double *tables);
int tphip_corrected_rates(tphip_plan *plan, const double *rates, const int32_t *nres, double *out);
int tphip_run_fused(tphip_plan *plan, const uint8_t *states, double *rate, double *subst, double *lnl,
                    uint8_t *flag, int32_t *nres, double *tables);
/* tphip_run_fused on a column range of a bigger taxon-major array: `states` points at the range's first column of taxon 0,
 * consecutive taxon rows are row_pitch bytes apart (0 = the plan's column count).  The upload is one 2-D copy; from pinned
 * memory no host copy is made.  With tphip_stage1_fit (opts.row_pitch) and tphip_plan_set_models this lets a caller stream a
 * big batch block by block -- stage 1, the per-site loop and PI of block k while the host still writes block k - 1's files. */
int tphip_run_fused_pitched(tphip_plan *plan, const uint8_t *states, int64_t row_pitch, double *rate, double *subst,
                            double *lnl, uint8_t *flag, int32_t *nres, double *tables);
int tphip_townsend_pi_dense(int32_t device, const double *rates, int64_t n, const double *times, int32_t n_times,
                            double *out);
int tphip_quad_townsend(int32_t device, const double *rates, int64_t n, double a, double b, int32_t integ_mode,
                        double *integral, double *abserr);
int tphip_state_histogram(int32_t device, const uint8_t *states, int64_t ncols_total, int32_t ntaxa,
                          const int64_t *locus_offsets, int64_t nloci, int64_t *hist);
/* Unique site patterns per locus (HyPhy: GetDataInfo(dupInfo, filteredData), models_and_rates.bf:960-963; its stage 1
 * sums pattern likelihoods weighted by counts and its stage 2 fits one rate per pattern).  Columns of a locus that are
 * identical after normalising the state masks (0 -> 15) collapse into one pattern.  Outputs: out_states
 * [ntaxa][*out_npatterns] taxon-major (the buffer must hold ntaxa * ncols_total bytes), out_offsets [nloci+1] pattern
 * offsets per locus, out_weight [*out_npatterns] column counts (buffer of ncols_total doubles), out_map [ncols_total]
 * column -> pattern index (may be NULL).  Patterns of a locus are in hash order, deterministic for given input.
 * Exact: equality is decided on the full columns, hashes only group candidates.  Plan-less, host pointers. */
int tphip_compress_columns(int32_t device, const uint8_t *states, int64_t ncols_total, int32_t ntaxa,
                           const int64_t *locus_offsets, int64_t nloci, uint8_t *out_states, int64_t *out_offsets,
                           double *out_weight, int64_t *out_map, int64_t *out_npatterns);
/* host-pointer twin of tphip_locus_loglik_dev; d_states_cache may keep the alignment on the device between calls:
 * pass the address of a NULL void* the first time and free it with tphip_free_device when done (NULL = copy the
 * alignment on every call) */
int tphip_locus_loglik(tphip_plan *plan, const uint8_t *states, void **d_states_cache, int64_t nvec,
                       const double *blen_vecs, int64_t ncand, const int32_t *cand_locus, const double *cand_exch,
                       const int32_t *cand_vec, const double *cand_scale, const int32_t *cand_pidx,
                       const double *cand_pfac, double *out);
/* host-pointer twin of tphip_locus_gradient_dev (dlogt and d2logt may be NULL) */
int tphip_locus_gradient(tphip_plan *plan, const uint8_t *states, void **d_states_cache, int64_t nvec,
                         const double *blen_vecs, int64_t ncand, const int32_t *cand_locus, const double *cand_exch,
                         const int32_t *cand_vec, const double *cand_scale, const int32_t *cand_pidx,
                         const double *cand_pfac, double *lnl, double *dexch, double *dlogt, double *sum_dlogt,
                         double *d2logt);
int tphip_free_device(tphip_plan *plan, void *d_ptr);
/* Column multiplicities [ncols] (host) for tphip_locus_loglik / tphip_locus_gradient of this plan: the plan's columns
 * are then site patterns (tphip_compress_columns) and every pattern's log-likelihood counts weight times.
 * NULL removes them.  The site-rate path ignores them. */
int tphip_plan_set_column_weights(tphip_plan *plan, const double *weights);

/* Diagnostic (tests): log L and its first two derivatives with respect to u = log(siteRate) for every
 * column at a caller-chosen u[ncols]; no classification, no optimiser.  Host pointers. */
int tphip_eval_columns(tphip_plan *plan, const uint8_t *states, const double *u, double *f, double *g, double *h);
/* device-pointer twin (full-size property tests: nothing crosses PCIe); enqueues on `stream`, does not synchronise */
int tphip_eval_columns_dev(tphip_plan *plan, const uint8_t *d_states, const double *d_u, double *d_f, double *d_g,
                           double *d_h, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* TPHIP_H */
