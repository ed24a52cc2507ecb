// locus_lik_params.hpp -- parameter blocks and launchers of the stage-1 kernels of locus_lik_kernel.hpp (eigenbasis value
// kernel, reverse-mode gradient kernel); the kernels themselves are compiled in locus_lik_launch.hip.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include "gtr_model.hpp"
#include "locus_lik_common.hpp"

namespace tphip {

struct LikParams {
    const uint8_t* states;         // [ntaxa][ncols_total]
    int64_t ncols_total;
    const int64_t* locus_offsets;  // [nloci+1]
    const double* col_weight;      // [ncols_total] multiplicity of each column (site-pattern counts), null = 1
    const LocusModel* models;      // [nloci] (only pi is used)
    const int4* lops;              // [nops] traversal program for these kernels: {code, taxon, node whose branch the op
                                   // climbs, tape slot (BRANCH / PUSH: written; POP_MUL: the PUSH it pops)} -- one 16-byte
                                   // scalar load per op, fetched one op ahead
    int32_t nops;
    int32_t ntaxa;
    int32_t stage_states;          // 1: the block's state masks are staged in LDS [ntaxa][block] before each sweep
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
    double* out;                   // [ncand * nsplit] sum over columns of log L (partials when nsplit > 1)
    // Column split: a candidate's columns are cut into nsplit slices (multiples of the block size), one work item
    // each, so that a handful of candidates on long loci still fills the device; slice sums are added up in
    // fixed order by split_sum_kernel.
    int32_t nsplit;
};

#ifndef TPHIP_GRAD_MIN_WAVES
#define TPHIP_GRAD_MIN_WAVES 2   // waves per SIMD the register allocation must allow (2: <= 256 VGPRs, no scratch)
#endif
constexpr int kGradBlock = 128;
constexpr int kGradWaves = kGradBlock / 64;
constexpr int kGradEF = 12;     // per node: e^{lam_k t}[4], F01 F02 F03 F12 F13 F23, t, pad
constexpr int kGradSlots = 4;   // LDS accumulator addresses per (wave, branch); fewer (GradParams.nslots) on big trees

struct GradParams {
    LikParams L;                // candidates as for locus_loglik_kernel (cand_pidx / cand_pfac are honoured too); L.out = lnL
    int32_t ntape;              // slots written by the forward sweep; the reverse sweep's adjoint stack follows them
    int64_t ncand;
    double* tape;               // [gridDim.x][ntape + stack_depth][4][kGradBlock]
    // outputs are indexed by work item = cand * nsplit + slice (partials when nsplit > 1)
    double* out_dexch;          // [items][6]   d lnL / d r  (AC, AG, AT, CG, CT, GT), branch lengths held fixed
    double* out_dlogt;          // [items][nnodes] d lnL / d log t_b, or null
    double* out_sum_dlogt;      // [items]
    int32_t nslots;             // 4, 2 or 1: accumulator addresses per (wave, branch) that fit the LDS budget of this tree
    double* out_d2logt;         // [items][nnodes] d2 lnL / d (log t_b)^2 with everything else fixed (the diagonal of the
                                // Hessian: preconditions the optimiser), or null
    const double* cand_eig;     // [ncand][36] eigen-systems computed beforehand by lik_eigen_kernel (one thread per candidate),
                                // or null: thread 0 of the workgroup diagonalises Q itself (~20 us per work item)
};

// host side (locus_lik_launch.hip)
hipError_t launch_locus_loglik_kernel(dim3 grid, size_t lds_bytes, hipStream_t st, const LikParams& L);
hipError_t launch_locus_grad_kernel(dim3 grid, size_t lds_bytes, hipStream_t st, const GradParams* d_params);
hipError_t launch_split_sum_kernel(hipStream_t st, const double* part, double* out, int64_t ncand, int nsplit, int width);
hipError_t locus_loglik_kernel_allow_lds(size_t lds_bytes);
hipError_t locus_grad_kernel_allow_lds(size_t lds_bytes);
hipError_t locus_grad_kernel_occupancy(size_t lds_bytes, int* blocks_per_cu);

}  // namespace tphip
