// tphip_internal.hpp -- what the translation units of the host side share: the plan object behind include/tphip.h,
// the error channel and small device-buffer helpers.  Not part of the C ABI.
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <string>
#include <vector>

#include "gtr_model.hpp"
#include "locus_lik_params.hpp"
#include "locus_value_params.hpp"
#include "locus_grad2_params.hpp"
#include "site_rate_params.hpp"
#include "tphip.h"
#include "tree_program.hpp"

using namespace tphip;

// thread-local message behind tphip_last_error() (defined in tphip.hip)
extern thread_local std::string g_tphip_err;

inline int fail(int code, const std::string& msg) {
    g_tphip_err = msg;
    return code;
}

#define HIP_TRY(expr)                                                                                      \
    do {                                                                                                   \
        hipError_t e_ = (expr);                                                                            \
        if (e_ != hipSuccess)                                                                              \
            return fail(TPHIP_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_));                 \
    } while (0)

constexpr int kProfileRing = 1024;
constexpr double kPiFloor = 1e-12;

template <typename T>
struct DevBuf {
    T* p = nullptr;
    size_t n = 0;
    hipError_t alloc(size_t count) {
        n = count;
        return hipMalloc((void**)&p, (count ? count : 1) * sizeof(T));
    }
    hipError_t upload(const std::vector<T>& h) {
        hipError_t e = alloc(h.size());
        if (e != hipSuccess) return e;
        if (h.empty()) return hipSuccess;
        return hipMemcpy(p, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice);
    }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
    }
};

inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

// small RAII set of device buffers for the host-pointer conveniences
struct Scratch {
    std::vector<void*> bufs;
    ~Scratch() { for (void* q : bufs) if (q) (void)hipFree(q); }
    template <typename T> T* get(size_t count) {
        void* q = nullptr;
        if (hipMalloc(&q, (count ? count : 1) * sizeof(T)) != hipSuccess) return nullptr;
        bufs.push_back(q);
        return (T*)q;
    }
};

struct tphip_host_buffers;
struct tphip_saved_desc;

struct tphip_plan {
    tphip_host_buffers* hostbuf = nullptr;   // device buffers, streams and events of the host-pointer entry points
    // host-pointer entry points on big batches: the loci are cut into `host_split` groups, each a plan of its own with its
    // own streams, so that the upload of group k + 1 runs under the kernels of group k (host_run)
    tphip_saved_desc* saved = nullptr;       // deep copy of the descriptor the plan was created from
    std::vector<tphip_plan*> parts;
    std::vector<int64_t> part_locus;         // [parts + 1] locus boundaries
    int32_t host_split = 1;
    bool is_part = false;
    bool last_run_in_parts = false;
    int32_t device = 0;
    int32_t ntaxa = 0;
    int64_t nloci = 0, ncols = 0;
    int32_t T = 0, n_t = 0, n_i = 0, integ_mode = 0, threshold = 0, round_decimals = -1, start_rule = 0;
    double correction = 1.0;
    TreeProgram prog;
    std::vector<int64_t> h_offsets;
    int64_t n_site_chunks = 0, n_pi_chunks = 0;
    int32_t site_chunk_cols = kSiteBlock;  // columns per site_rate_kernel work slice (multiple of 64)
    DevBuf<TreeOp> d_ops, d_fused_ops;
    DevBuf<LocusModel> d_models;
    DevBuf<int64_t> d_offsets, d_locus_pichunk_offsets;
    DevBuf<int32_t> d_tip_taxon, d_op_node;
    // launch configuration of the locus likelihood / gradient kernels, fixed at plan creation (tuning knobs are read
    // from the environment once, there): LDS bytes, whether the gradient kernel stages its state masks, resident
    // gradient blocks per CU, forced slice count (0 = automatic)
    size_t lik_lds = 0, grad_lds = 0;
    int32_t lik_stage = 0, grad_stage = 0, grad_blocks_per_cu = 1, lik_nsplit_forced = 0, grad_slots = 4;
    bool lik_ok = false, grad_ok = false;
    int32_t ncat = 0;
    DevBuf<double> d_cat;     // [2 * ncat] category rates, then log weights
    DevBuf<int4> d_lik_ops;   // {code, taxon, node, tape slot} per op for the locus likelihood / gradient kernels
    double* d_tape = nullptr;   // reverse-mode tape of locus_grad_kernel, grown on demand
    double* d_grad_eig = nullptr;   // per-candidate eigen-systems for locus_grad_kernel
    size_t grad_eig_bytes = 0;
    double* d_value_ws = nullptr;   // per-candidate eigen-systems + transition matrices of locus_value_kernel (one chunk)
    size_t value_ws_bytes = 0;
    const uint8_t* lib_states = nullptr;   // the device copy of the alignment the library made itself (stage_alignment) ...
    uint32_t* d_value_packed = nullptr;    // ... and its state codes packed for locus_value_kernel (valid for exactly that copy)
    int32_t value_cols = 0;         // columns per thread of locus_value_kernel (0: tree too large for it, eigenbasis kernel instead)
    size_t value_lds = 0;
    DevBuf<int4> d_value_ops;       // fused op stream of locus_value_kernel
    DevBuf<int32_t> d_value_tip_node;
    int32_t value_nops = 0;
    size_t tape_bytes = 0;
    // grow-only device arena + pinned host mirror for the host-pointer likelihood / gradient calls: the optimiser
    // makes thousands of small calls, so they must not hipMalloc or issue a dozen pageable copies each
    char* d_arena = nullptr;
    char* h_arena = nullptr;
    size_t arena_bytes = 0;
    void* d_grad_params = nullptr;   // device copy of the gradient kernel's parameter block
    // transition-matrix gradient kernel (locus_grad2_kernel.hpp): binary trees with the value kernel's packed state codes
    bool grad2_ok = false;
    DevBuf<int4> d_grad2_fops, d_grad2_rops;
    int32_t grad2_nrops = 0, grad2_ntape = 0, grad2_rdepth = 0, grad2_blocks_per_cu = 1;
    size_t grad2_lds = 0;
    double* d_grad2_ws = nullptr;    // per-candidate transition matrices + branch tables of one chunk of candidates
    size_t grad2_ws_bytes = 0;
    void* d_grad2_params = nullptr;
    double* d_col_weight = nullptr;  // optional column multiplicities for the locus likelihood / gradient kernels
    double* d_part = nullptr;   // per-slice partial sums of the locus likelihood / gradient kernels, grown on demand
    size_t part_bytes = 0;
    int64_t max_locus_cols = 0;
    int32_t nnodes = 0;
    int32_t nwords = 0;
    DevBuf<int32_t> d_site_chunk_locus, d_site_chunk_index, d_pi_chunk_locus, d_pi_chunk_index, d_times, d_intervals;
    DevBuf<unsigned long long> d_evals;
    // workspace layout (bytes)
    size_t ws_work_cols2 = 0, ws_work_cols = 0, ws_work_count = 0, ws_work_prefix = 0, ws_slice_prefix = 0, ws_partial = 0, ws_packed = 0, ws_total = 0;
    size_t ws_hash = 0, ws_dup_of = 0, ws_tab_key = 0, ws_tab_val = 0, ws_dedup_on = 0;   // site-pattern de-duplication
    int32_t dedup_mode = 0;   // DEDUP_AUTO (pattern_kernels.hpp)
    int32_t num_cus = 256;
    int32_t site_waves = 0;  // persistent grid of site_rate_kernel = resident waves on the device
    int32_t site_persistent = 1;
    int32_t mixed_few_waves = 0;   // mixed-loci mode: workgroups that take shares when the work list is short (SiteParams)
    int64_t mixed_switch_cols = 0;
    int32_t site_mixed = 0;        // small batches: waves carry columns of several loci (site_rate_kernel<NW, false, true>)
    int32_t site_grid_mult = 1;   // persistent grid = resident waves x this (see plan creation)
    int32_t site_lds_depth = 0;   // parked partials kept in LDS by site_rate_kernel (< stack depth: SPILL variant)
    size_t ws_spill = 0;
    double site_first_fraction = 0.0;   // share of the work the first round of shares takes (0 = equal shares)
    bool force_byte_path = false;       // TPHIP_FORCE_BYTE_PATH=1 at plan creation: run the NW = 0 kernel on any tree (tests)
    // profiling
    bool profile = false;
    std::vector<hipEvent_t> ev;  // 4 events per slot: site start/stop, pi start/stop
    int ev_used = 0;
    double acc_site_ms = 0, acc_pi_ms = 0;
    int64_t acc_launches = 0;
};

// helpers of tphip.hip for the other translation units of the library (not declared in include/tphip.h)
// (hidden visibility: the shared library exports exactly what include/tphip.h declares)
extern "C" {
__attribute__((visibility("hidden"))) int tphip_internal_stage_alignment(tphip_plan* p, const uint8_t* states, void** d_states_cache,
                                                                       uint8_t** d_s);
__attribute__((visibility("hidden"))) const tphip_plan_desc* tphip_internal_saved_desc(const tphip_plan* p);
__attribute__((visibility("hidden"))) int tphip_internal_compress_dev(const uint8_t* d_s, int64_t ncols_total, int32_t ntaxa,
                                                                    const int64_t* d_off, int64_t nloci, uint8_t** d_out,
                                                                    int64_t** d_newoff, double** d_w, int64_t* npat);
__attribute__((visibility("hidden"))) int tphip_internal_store_pi(tphip_plan* p, const double* pi);
__attribute__((visibility("hidden"))) int tphip_internal_set_models_dev(tphip_plan* p, const double* d_pi, const double* d_exch,
                                                                      void* stream);
}
