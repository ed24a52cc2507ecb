// site_rate_params.hpp -- launch parameters and constants of the site-rate kernels, shared by the two translation
// units of libtphip.so: site_rate_launch.hip (compiles site_rate_kernel.hpp, the only code built with
// -structurizecfg-skip-uniform-regions) and tphip.hip (everything else, built without that flag).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include "gtr_model.hpp"
#include "tree_program.hpp"

namespace tphip {

constexpr int kFirstStepMinTaxa = 32;   // trees from this size on use the parsimony length in the optimiser's first step (pi_kernels.hpp)
constexpr double kUMin = -23.025850929940457;  // log(1e-10)
constexpr double kUMax = 9.210340371976184;    // log(1e4)
constexpr double kStepMax = 2.0;
constexpr double kStepTol = 3e-4;       // accept when the step is this small: applied with a third-order correction
constexpr double kHermiteSpan = 3e-4;   // ... while |step| * |distance between the two points| stays below this
constexpr double kHermiteTol = 2e-3;    // final step from the two-point quartic model of f' accepted below this size
                                        // (calibration of the two bounds: tools/debug/two_point_experiment.py)
constexpr double kHermiteT2D3 = 1.5e-8;  // step^2 * |distance of the far point|^3 below this (the quartic's error term)
constexpr double kHermiteGuard = 0.005;  // the quartic's higher-order terms at the step, relative to |h|
constexpr double kHermiteRegular = 0.25; // curvature |h| from which the two exits apply (below: iterate until the step is below 1e-6)
constexpr double kHermiteNoise = 5.3e-5; // 30 * (8 * 2.2e-16) / 1e-9: rounding of the two values vs curvature (site_rate_kernel.hpp)
constexpr double kHalleySpan = 1.0;      // the two curvatures behind the Halley step of a weakly curved point lie this close
constexpr double kStepTolFirst = 1e-6;  // ... except at the first evaluation (no second point yet)
constexpr int kMaxIt = 100;
constexpr double kFlatEps = 1e-10;  // |g| and |h| below this: log L flat to fp64 resolution -> saturated
// A maximum found at a large rate may be nothing but rounding noise on the plateau log L reaches as s -> infinity (where
// the flatness rule above fires, or not, depending on the last bits).  So an optimum at s >= kCheckRate is confirmed
// by value: if log L at the largest rate is not lower by more than kSatTol (relative), the column is saturated.
constexpr double kUCheck = 2.995732273553991;   // log(20)
constexpr double kSatTol = 1e-10;
constexpr double kPlateauStride = 0.5;
constexpr int kSiteBlock = 64;      // one wavefront per workgroup
constexpr int kMixedLoci = 8;        // loci whose columns a wave of the mixed-loci mode carries at a time (site_rate_kernel.hpp)
constexpr int kMixedLdsHeader = kMixedLoci * 64 + 64 + 32;   // doubles: tip tables, 2^(j/64), segment table
constexpr int kMixedColBits = 28;    // a work entry of that mode = column | locus-in-group << 28: batches below 2^28 columns
constexpr int kSiteLdsHeader = 160;  // doubles of LDS before the stack: tip table [16][4] + model [32] + 2^(j/64) [64]
#ifndef TPHIP_EXP_TABLE
#define TPHIP_EXP_TABLE 1
#endif
constexpr bool kUseExpTable = TPHIP_EXP_TABLE != 0;

struct SiteParams {
    const uint8_t* states;       // [ntaxa][ncols_total]
    int64_t ncols_total;
    const LocusModel* models;    // [nloci]
    const TreeOp* ops;
    int32_t nops;
    int32_t stack_depth;
    double chrono_length;
    const int64_t* locus_offsets;  // [nloci+1]
    const int32_t* chunk_locus;    // [nchunks]  work slices of chunk_cols columns, never straddling loci
    const int32_t* chunk_index;    // [nchunks]  index of the slice inside its locus
    int32_t chunk_cols;            // target columns per slice
    const uint32_t* packed;        // [nwords][ncols_total] 8 tip masks per word, tips in program order (classify_kernel)
    int32_t nwords;                // ceil(ntips / 8)
    const int32_t* work_cols;      // [ncols_total] compacted column ids, per locus at locus_offsets[l]
    int32_t* work_cols2;           // [ncols_total] scratch of the same shape, or null: in the small-batch mode every wave
                                   // rewrites its slice of the list there with the columns predicted to be slow first
    const int32_t* work_count;     // [nloci]
    const int64_t* work_prefix;    // [nloci+1] exclusive scan of work_count (scan_counts_kernel)
    const int64_t* slice_prefix;   // [nloci+1] exclusive scan of the per-locus slice counts (non-persistent mode)
    int64_t nloci;
    int32_t persistent;            // 1: grid = resident waves (x grid multiplier), shares of the global work list; 0: grid = slices
                                   // (the mixed-loci variants are launched with 1 and equal shares)
    int32_t first_round;           // persistent: the first `first_round` workgroups (one per resident wave) share
    double first_fraction;         //   this fraction of the work equally, the others the rest (see site_rate_kernel)
    int32_t mixed_few_waves;       // mixed-loci mode: with fewer than mixed_switch_cols columns on the work list only this many
    int64_t mixed_switch_cols;     //   workgroups take shares (one wave per SIMD), the others leave at once
    int32_t ncat;                  // > 1: discrete rate mixture on top of the site rate (tphip_plan_desc.ncat)
    const double* cat;             // [2 * ncat]: category rate multipliers, then log weights
    double* rate;
    double* subst;
    double* lnl;
    uint8_t* flag;
    unsigned long long* eval_counter;
    double* spill;                 // SPILL variant: [grid waves][stack_depth - lds_depth][12][64] parked partials beyond the LDS stack
    int32_t lds_depth;             // parked partials kept in LDS (= stack_depth unless the SPILL variant runs)
};

// Slices a locus with `count` optimiser columns is cut into in the non-persistent mode: about chunk_cols each.
__host__ __device__ __forceinline__ int site_slices(int count, int chunk_cols) {
    if (count <= 0) return 0;
    const int ns = (count + chunk_cols / 2) / chunk_cols;
    return ns < 1 ? 1 : ns;
}

constexpr int kStreamWords = -1;   // NW value of the streamed-words path
constexpr int kStreamWordsSpill = -2;   // launcher variant: streamed words + SPILL
constexpr int kMixedVariant = 100;   // launcher variants 102 / 108: packed words in registers, mixed-loci mode

// Diagnostic launch: evaluate f = log L, g = df/du, h = d2f/du2 at a caller-chosen u for EVERY column.
struct EvalParams {
    SiteParams S;
    const double* u;
    double* f;
    double* g;
    double* h;
};

// ---- launchers defined in site_rate_launch.hip ------------------------------------------------------------------
// variant: 0 = byte path (NW = 0), 2 / 8 = packed words in registers, kStreamWords = streamed words (> 64 tips),
// kStreamWordsSpill = the same with the deepest parked partials in global scratch (S.spill, S.lds_depth)
hipError_t launch_site_rate_kernel(int variant, dim3 grid, size_t lds_bytes, hipStream_t st, const SiteParams& S);
hipError_t site_rate_kernel_occupancy(int variant, size_t lds_bytes, int* blocks_per_cu);
hipError_t launch_eval_columns_kernel(dim3 grid, size_t lds_bytes, hipStream_t st, const EvalParams& E);
hipError_t launch_scan_counts_kernel(hipStream_t st, const int32_t* count, int64_t nloci, int32_t chunk_cols, int64_t* prefix,
                                     int64_t* slice_prefix);

}  // namespace tphip
