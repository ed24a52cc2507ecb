// stage1_driver.hip -- host side of HyPhy's stage 1 inside the engine (tphip_stage1_fit / tphip_stage1_fit_dev).
//
// What it replaces, per locus, for all loci of a plan at once (tapir/data/models_and_rates.bf):
//   bf:487-520   general reversible model: Optimize over the 5 exchangeabilities (AG = 1) AND every branch length
//   bf:522-540   stash the fitted branch lengths in expected substitutions (t_b * totalFactor)
//   bf:542-661   the other 202 partitions of the six rates into classes, lengths = stash / the model's own totalFactor
//   bf:806-847   Akaike weights w_m ~ exp(lnL_m - k_m) and the model-averaged AC, AT, CG, CT, GT handed to stage 2
//
// The host only sequences launches: the optimisers' state lives in HBM and every step of them is a kernel of
// stage1_opt_kernels.hpp; likelihoods and gradients are the engine's own kernels (tphip_locus_loglik_dev,
// tphip_locus_gradient_dev).  Per optimiser step the host reads ONE 32-byte counter block (how many candidates to launch).
// Round 2 ran this loop in Python on torch tensors: ~10^5 small library launches per pass over 2000 loci.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include <hipcub/hipcub.hpp>

#include "stage1_opt_kernels.hpp"
#include "tphip_internal.hpp"

using namespace tphip;
using namespace tphip::s1;


namespace {

// the 203 rate-class models as restricted growth strings over (AC, AG, AT, CG, CT, GT): '012345' first (the general model,
// bf:487), then the other 202 in the order of the loops at bf:544-566, which is the lexicographic order of the strings
void enumerate_models(std::vector<std::string>* out) {
    out->clear();
    out->push_back("012345");
    std::string s = "0";
    struct Rec {
        static void go(std::string& cur, int maxc, std::vector<std::string>* o) {
            if (cur.size() == 6) { if (cur != "012345") o->push_back(cur); return; }
            for (int c = 0; c <= maxc + 1; ++c) {
                cur.push_back((char)('0' + c));
                go(cur, std::max(maxc, c), o);
                cur.pop_back();
            }
        }
    };
    Rec::go(s, 0, out);
}

// class of each rate with the class of AG mapped to -1 (fixed at 1, bf:577-600) and the free classes renumbered 0..k-1
void model_design(const std::vector<std::string>& strings, std::vector<int8_t>* cls, std::vector<int32_t>* kk) {
    cls->assign(strings.size() * 6, 0);
    kk->assign(strings.size(), 0);
    for (size_t m = 0; m < strings.size(); ++m) {
        const std::string& s = strings[m];
        const char ag = s[1];
        std::string free_;
        for (char c : s)
            if (c != ag && free_.find(c) == std::string::npos) free_.push_back(c);
        (*kk)[m] = (int32_t)free_.size();
        for (int q = 0; q < 6; ++q) (*cls)[m * 6 + q] = s[q] == ag ? (int8_t)-1 : (int8_t)free_.find(s[q]);
    }
}

// bump allocator over one device allocation; first pass (base == nullptr) only measures
struct Arena {
    char* base = nullptr;
    size_t off = 0;
    template <typename T> T* take(size_t n) {
        const size_t o = off;
        off = align_up(off + std::max<size_t>(n, 1) * sizeof(T), 256);
        return base ? (T*)(base + o) : nullptr;
    }
};

struct Driver {
    tphip_plan* p;
    const uint8_t* d_states;
    hipStream_t st;
    int32_t* d_counters;
    int32_t* h_counters;   // pinned
    int64_t nevals = 0, ngrads = 0, nlaunch = 0, nsync = 0;

    int read_counters() {
        HIP_TRY(hipMemcpyAsync(h_counters, d_counters, sizeof(int32_t) * C_COUNT, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
        ++nsync;
        return TPHIP_OK;
    }
    int zero(int first, int count) {
        HIP_TRY(hipMemsetAsync(d_counters + first, 0, sizeof(int32_t) * count, st));
        return TPHIP_OK;
    }
    int set_counter(int which, int32_t v) {
        h_counters[C_COUNT + which] = v;   // second half of the pinned block is the upload staging
        HIP_TRY(hipMemcpyAsync(d_counters + which, h_counters + C_COUNT + which, sizeof(int32_t), hipMemcpyHostToDevice, st));
        return TPHIP_OK;
    }
    int value(const CandArrays& C, const double* vecs, int64_t n) {
        if (n <= 0) return TPHIP_OK;
        nevals += n;
        return tphip_locus_loglik_dev(p, d_states, n, C.locus, C.exch, vecs, C.vec, C.scale, C.pidx, C.pfac, C.out, (void*)st);
    }
};

CandArrays take_cands(Arena& A, size_t cap) {
    CandArrays C;
    C.locus = A.take<int32_t>(cap); C.exch = A.take<double>(cap * 6); C.vec = A.take<int32_t>(cap);
    C.scale = A.take<double>(cap); C.pidx = A.take<int32_t>(cap); C.pfac = A.take<double>(cap);
    C.prob = A.take<int32_t>(cap); C.out = A.take<double>(cap);
    return C;
}

#define RC(expr) do { int rc_ = (expr); if (rc_) return rc_; } while (0)
#define KCHECK() HIP_TRY(hipGetLastError())

unsigned blocks_for(int64_t n, int b) { return (unsigned)std::max<int64_t>(1, (n + b - 1) / b); }

}  // namespace

// The plan seen through other columns for the duration of a stage-1 call: the site patterns of its alignment instead of the
// raw columns (opts.compress_patterns).  The likelihood entry points read these fields of the plan; the destructor puts the
// plan's own back.
struct PatternView {
    tphip_plan* p = nullptr;
    int64_t ncols = 0, max_locus_cols = 0;
    int64_t* d_offsets = nullptr;
    double* d_col_weight = nullptr;
    const uint8_t* lib_states = nullptr;
    uint32_t* d_value_packed = nullptr;
    std::vector<int64_t> h_offsets;
    uint8_t* d_pat = nullptr; int64_t* d_poff = nullptr; double* d_pw = nullptr; uint32_t* d_ppacked = nullptr;
    void install(tphip_plan* plan, uint8_t* pat, int64_t* poff, double* pw, uint32_t* packed, int64_t npat, std::vector<int64_t>&& hoff) {
        p = plan;
        ncols = p->ncols; max_locus_cols = p->max_locus_cols; d_offsets = p->d_offsets.p; d_col_weight = p->d_col_weight;
        lib_states = p->lib_states; d_value_packed = p->d_value_packed; h_offsets.swap(p->h_offsets);
        d_pat = pat; d_poff = poff; d_pw = pw; d_ppacked = packed;
        p->ncols = npat; p->d_offsets.p = poff; p->d_col_weight = pw; p->lib_states = packed ? pat : nullptr; p->d_value_packed = packed;
        p->h_offsets = std::move(hoff);
        p->max_locus_cols = 0;
        for (size_t l = 0; l + 1 < p->h_offsets.size(); ++l)
            p->max_locus_cols = std::max<int64_t>(p->max_locus_cols, p->h_offsets[l + 1] - p->h_offsets[l]);
    }
    ~PatternView() {
        if (p) {
            p->ncols = ncols; p->max_locus_cols = max_locus_cols; p->d_offsets.p = d_offsets; p->d_col_weight = d_col_weight;
            p->lib_states = lib_states; p->d_value_packed = d_value_packed; p->h_offsets.swap(h_offsets);
        }
        for (void* q : {(void*)d_pat, (void*)d_poff, (void*)d_pw, (void*)d_ppacked}) if (q) (void)hipFree(q);
    }
};

extern "C" int tphip_stage1_fit_dev(tphip_plan* p, const uint8_t* d_states, const tphip_stage1_opts* opts_in, double* exch_out,
                                    double* pi_out, double* weights_out, double* lnl_out, double* model_exch_out, double* grm_blen_out,
                                    int32_t* grm_iters_out, int32_t* sub_iters_out, int64_t* stats_out, void* stream) {
    if (!p || !d_states || !exch_out) return fail(TPHIP_ERR_INVALID, "null argument");
    tphip_stage1_opts opt;
    memset(&opt, 0, sizeof(opt));
    if (opts_in) {
        if (opts_in->struct_size < sizeof(uint32_t) || opts_in->struct_size > 4096)
            return fail(TPHIP_ERR_INVALID, "tphip_stage1_opts.struct_size not set");
        memcpy(&opt, opts_in, std::min<size_t>(opts_in->struct_size, sizeof(opt)));
    }
    const tphip_plan_desc* desc = tphip_internal_saved_desc(p);
    if (!desc) return fail(TPHIP_ERR_INVALID, "plan has no saved descriptor");
    HIP_TRY(hipSetDevice(p->device));
    hipStream_t st = (hipStream_t)stream;
    const int L = (int)p->nloci, nn = p->nnodes;
    if (L < 1) return fail(TPHIP_ERR_INVALID, "plan without loci");
    // branch coordinates: every node but the root, in node order; the two branches below a bifurcating root share one
    // coordinate at the input tree's proportion (the likelihood of a reversible model sees only their sum: GrmState)
    std::vector<int32_t> branches, partner, node_coord(nn, -1);
    std::vector<double> node_w(nn, 1.0);
    {
        const int root = nn - 1;
        std::vector<int> rk;
        for (int n = 0; n < nn; ++n) if (desc->parent[n] == root) rk.push_back(n);
        const bool merge = rk.size() == 2 && !opt.free_root_pair;
        for (int n = 0; n < nn; ++n) {
            if (desc->parent[n] < 0) continue;
            if (merge && n == rk[1]) {
                node_coord[n] = node_coord[rk[0]];
                partner[node_coord[n]] = n;
                const double a = std::max(desc->branch_len[rk[0]], 0.0), b = std::max(desc->branch_len[rk[1]], 0.0);
                const double w0 = (a + b > 0) ? std::min(std::max(a / (a + b), 0.05), 0.95) : 0.5;
                node_w[rk[0]] = w0; node_w[n] = 1.0 - w0;
                continue;
            }
            node_coord[n] = (int32_t)branches.size();
            branches.push_back(n);
            partner.push_back(-1);
        }
    }
    const int nb = (int)branches.size(), D = 5 + nb;
    const int maxit_grm = opt.maxit_grm > 0 ? opt.maxit_grm : std::max(300, 4 * D);
    const int maxit_sub = opt.maxit_sub > 0 ? opt.maxit_sub : 100;
    const double fd_step = opt.fd_step > 0 ? opt.fd_step : 1e-4;
    const int prune = opt.no_prune ? 0 : 1;
    // model tables
    std::vector<std::string> strings;
    enumerate_models(&strings);
    if ((int)strings.size() != kModels) return fail(TPHIP_ERR_INVALID, "internal error: model enumeration");
    std::vector<int8_t> cls_all;
    std::vector<int32_t> kk_all;
    model_design(strings, &cls_all, &kk_all);
    // start shape of the branch lengths: the input tree's lengths over their mean, floored
    std::vector<double> h_shape(nn, 0.0);
    {
        double mean = 0;
        for (int n : branches) mean += std::max(desc->branch_len[n], 0.0);
        mean = std::max(mean / std::max(nb, 1), 1e-300);
        for (int n : branches) h_shape[n] = std::max(std::max(desc->branch_len[n], 0.0) / mean, 1e-3);
    }
    const int ngrid = 17;
    std::vector<double> h_grid(ngrid);
    for (int k = 0; k < ngrid; ++k) h_grid[k] = std::pow(10.0, -4.0 + 4.0 * k / (ngrid - 1));

    // ---- device memory: one allocation, laid out twice (measure, then place) ------------------------------------------------
    const size_t P = (size_t)L, LM = (size_t)L * kSubModels;
    const size_t cap = std::max<size_t>({P * ngrid, P * kScreenPoints, LM * 9});
    GrmState G;
    ScreenState S;
    SubState B;
    struct Extra {
        int32_t* counters; int32_t *branches, *node_coord, *partner; double *dk, *shape, *grid, *node_w; int8_t* cls; int32_t* kk;
        int32_t *parent, *leaf; double* pchanges;
        double *grm_exch, *grm_blen, *stash, *grm_lnl; uint8_t* flags; int32_t* sub_iters; void* cub_tmp; int32_t* nsel;
        double *o_weights, *o_lnl, *o_mexch, *o_exch, *pi; unsigned long long* hist;
    } X;
    size_t cub_bytes = 0;
    {
        hipcub::CountingInputIterator<int32_t> it0(0);
        HIP_TRY(hipcub::DeviceSelect::Flagged(nullptr, cub_bytes, it0, (uint8_t*)nullptr, (int32_t*)nullptr, (int32_t*)nullptr, (int)LM, st));
    }
    auto layout = [&](Arena& A) {
        X.counters = A.take<int32_t>(C_COUNT);
        X.branches = A.take<int32_t>(nb); X.node_coord = A.take<int32_t>(nn); X.partner = A.take<int32_t>(nb);
        X.node_w = A.take<double>(nn); X.dk = A.take<double>(P * 6);
        X.parent = A.take<int32_t>(nn); X.leaf = A.take<int32_t>(nn); X.pchanges = A.take<double>(P * (nn + 1));
        X.shape = A.take<double>(nn); X.grid = A.take<double>(ngrid); X.cls = A.take<int8_t>(kSubModels * 6); X.kk = A.take<int32_t>(kSubModels);
        X.grm_exch = A.take<double>(P * 6); X.grm_blen = A.take<double>(P * nn); X.stash = A.take<double>(P * nn); X.grm_lnl = A.take<double>(P);
        X.flags = A.take<uint8_t>(LM); X.sub_iters = A.take<int32_t>(LM); X.cub_tmp = A.take<char>(cub_bytes); X.nsel = A.take<int32_t>(1);
        X.o_weights = A.take<double>(P * kModels); X.o_lnl = A.take<double>(P * kModels); X.o_mexch = A.take<double>(P * kModels * 6);
        X.o_exch = A.take<double>(P * 6); X.pi = A.take<double>(P * 4); X.hist = A.take<unsigned long long>(P * 16);
        CandArrays C = take_cands(A, cap);
        // general model
        G.P = L; G.D = D; G.nb = nb; G.nn = nn; G.bspace_metric = getenv("TPHIP_S1_LOGMETRIC") ? 0 : 1; G.node_coord = X.node_coord; G.branches = X.branches; G.dk = X.dk;
        G.node_w = X.node_w; G.partner = X.partner;
        G.x = A.take<double>(P * D); G.g = A.take<double>(P * D); G.hd = A.take<double>(P * D); G.d = A.take<double>(P * D);
        G.xt = A.take<double>(P * D); G.S = A.take<double>((size_t)kHistory * P * D); G.Y = A.take<double>((size_t)kHistory * P * D);
        G.rho = A.take<double>((size_t)kHistory * P);
        G.f = A.take<double>(P); G.fnew = A.take<double>(P); G.t = A.take<double>(P); G.gd = A.take<double>(P);
        G.gamma = A.take<double>(P); G.last_df = A.take<double>(P);
        G.nhist = A.take<int32_t>(P); G.head = A.take<int32_t>(P); G.kicks = A.take<int32_t>(P); G.iters = A.take<int32_t>(P);
        G.ls_round = A.take<int32_t>(P); G.slot = A.take<int32_t>(P);
        G.phase = A.take<uint8_t>(P); G.fresh = A.take<uint8_t>(P);
        G.counters = X.counters; G.C = C;
        G.vecs = A.take<double>(P * nn);
        G.o_lnl = A.take<double>(P); G.o_dexch = A.take<double>(P * 6); G.o_dlogt = A.take<double>(P * nn);
        G.o_sdl = A.take<double>(P); G.o_d2 = A.take<double>(P * nn);
        // screen
        S.L = L; S.grm_exch = X.grm_exch; S.grm_lnl = X.grm_lnl; S.w6 = X.dk; S.cls = X.cls; S.kk = X.kk;
        S.K = A.take<double>(P * 25); S.g5 = A.take<double>(P * 5); S.theta = A.take<double>(LM * 4); S.hdiag = A.take<double>(LM * 4);
        S.f_at = A.take<double>(LM); S.best = A.take<double>(P); S.keep = A.take<int32_t>(LM); S.counters = X.counters; S.C = C;
        // rate-class models (at most LM problems)
        B.id = S.keep; B.cls = X.cls; B.kk = X.kk; B.w6 = X.dk; B.h = fd_step;
        B.x = A.take<double>(LM * 4); B.g = A.take<double>(LM * 4); B.d = A.take<double>(LM * 4); B.xt = A.take<double>(LM * 4);
        B.Hinv = A.take<double>(LM * 16); B.H0 = A.take<double>(LM * 16); B.f_known = S.f_at;
        B.f = A.take<double>(LM); B.fnew = A.take<double>(LM); B.t = A.take<double>(LM); B.gd = A.take<double>(LM);
        B.last_df = A.take<double>(LM); B.df = A.take<double>(LM);
        B.kicks = A.take<int32_t>(LM); B.iters = A.take<int32_t>(LM); B.ls_round = A.take<int32_t>(LM); B.slot = A.take<int32_t>(LM);
        B.phase = A.take<uint8_t>(LM); B.best = S.best; B.counters = X.counters; B.C = C;
    };
    Arena A0;
    layout(A0);
    char* d_base = nullptr;
    HIP_TRY(hipMalloc((void**)&d_base, A0.off));
    int32_t* h_counters = nullptr;
    if (hipHostMalloc((void**)&h_counters, sizeof(int32_t) * 2 * C_COUNT, hipHostMallocDefault) != hipSuccess) {
        (void)hipFree(d_base);
        return fail(TPHIP_ERR_HIP, "hipHostMalloc failed");
    }
    struct Cleanup {
        char* d; int32_t* h;
        ~Cleanup() { if (d) (void)hipFree(d); if (h) (void)hipHostFree(h); }
    } cleanup{d_base, h_counters};
    Arena A1;
    A1.base = d_base;
    layout(A1);
    Driver dr{p, d_states, st, X.counters, h_counters};

    // constants up
    std::vector<int8_t> h_cls(cls_all.begin() + 6, cls_all.end());
    std::vector<int32_t> h_kk(kk_all.begin() + 1, kk_all.end());
    HIP_TRY(hipMemcpyAsync(X.branches, branches.data(), sizeof(int32_t) * nb, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(X.node_coord, node_coord.data(), sizeof(int32_t) * nn, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(X.partner, partner.data(), sizeof(int32_t) * nb, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(X.node_w, node_w.data(), sizeof(double) * nn, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(X.shape, h_shape.data(), sizeof(double) * nn, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(X.parent, desc->parent, sizeof(int32_t) * nn, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(X.leaf, desc->leaf_taxon, sizeof(int32_t) * nn, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(X.grid, h_grid.data(), sizeof(double) * ngrid, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(X.cls, h_cls.data(), sizeof(int8_t) * kSubModels * 6, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(X.kk, h_kk.data(), sizeof(int32_t) * kSubModels, hipMemcpyHostToDevice, st));
    HIP_TRY(hipStreamSynchronize(st));   // the host vectors above go out of use

    // ---- base frequencies and site patterns (bf:960-968) ---------------------------------------------------------------------
    if (opt.empirical_pi) {
        int rc = tphip_state_histogram_dev(p->device, d_states, p->ncols, p->ntaxa, p->d_offsets.p, L, (int64_t*)X.hist, (void*)st);
        if (rc) return rc;
        hipLaunchKernelGGL(empirical_pi_kernel, dim3(blocks_for(L, 64)), dim3(64), 0, st, L, (const unsigned long long*)X.hist, X.pi);
        KCHECK();
        std::vector<double> ones((size_t)L * 6, 1.0);
        HIP_TRY(hipMemcpyAsync(X.o_mexch, ones.data(), sizeof(double) * L * 6, hipMemcpyHostToDevice, st));   // (a free buffer)
        RC(tphip_internal_set_models_dev(p, X.pi, X.o_mexch, (void*)st));
        std::vector<double> hpi((size_t)L * 4);
        HIP_TRY(hipMemcpyAsync(hpi.data(), X.pi, sizeof(double) * L * 4, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
        RC(tphip_internal_store_pi(p, hpi.data()));
        desc = tphip_internal_saved_desc(p);
    }
    PatternView view;
    if (opt.compress_patterns && p->ncols > 0) {
        HIP_TRY(hipStreamSynchronize(st));
        uint8_t* d_pat = nullptr; int64_t* d_poff = nullptr; double* d_pw = nullptr; int64_t npat = 0;
        RC(tphip_internal_compress_dev(d_states, p->ncols, p->ntaxa, p->d_offsets.p, L, &d_pat, &d_poff, &d_pw, &npat));
        std::vector<int64_t> hoff((size_t)L + 1);
        uint32_t* d_packed = nullptr;
        hipError_t e = hipMemcpy(hoff.data(), d_poff, sizeof(int64_t) * ((size_t)L + 1), hipMemcpyDeviceToHost);
        if (e == hipSuccess && p->value_cols > 0 && npat > 0) {
            e = hipMalloc((void**)&d_packed, sizeof(uint32_t) * (size_t)p->nwords * (size_t)npat);
            if (e == hipSuccess) e = launch_value_pack_codes_kernel(st, d_pat, npat, p->d_tip_taxon.p, p->nwords, d_packed);
        }
        if (e != hipSuccess) {
            for (void* q : {(void*)d_pat, (void*)d_poff, (void*)d_pw, (void*)d_packed}) if (q) (void)hipFree(q);
            return fail(TPHIP_ERR_HIP, std::string("site patterns: ") + hipGetErrorString(e));
        }
        view.install(p, d_pat, d_poff, d_pw, d_packed, npat, std::move(hoff));
        d_states = d_pat;
        dr.d_states = d_pat;
    }
    hipLaunchKernelGGL(dk_kernel, dim3(blocks_for(L, 64)), dim3(64), 0, st, L, p->d_models.p, X.dk);
    KCHECK();

    // ---- general model ---------------------------------------------------------------------------------------------------------
    // start: tree shape times the best of a grid of scales (17 likelihoods per locus under the all-ones model)
    hipLaunchKernelGGL(grm_grid_emit_kernel, dim3(blocks_for((int64_t)L * ngrid, 256)), dim3(256), 0, st, G.C, L, ngrid, X.grid);
    KCHECK();
    RC(dr.value(G.C, X.shape, (int64_t)L * ngrid));
    HIP_TRY(hipMemsetAsync(X.counters + C_PARS, 0, sizeof(int32_t), st));
    hipLaunchKernelGGL(grm_grid_pick_kernel, dim3(blocks_for(L, 256)), dim3(256), 0, st, G, ngrid, X.grid, X.shape);
    KCHECK();
    // Second start: per-branch parsimony counts shrunk towards the grid start (one more likelihood per locus); the better of
    // the two is kept.  The input tree's shape is what the reference hands HyPhy, and on loci that follow it the grid start is
    // hard to beat (21.2 iterations against 22.0); but where the input lengths are off -- each branch by a factor exp(N(0, 2)):
    // 36 iterations from the grid start, and 472 of 500 loci end hundreds of log-units lower with a few branches parked at
    // saturating lengths, where the gradient vanishes -- the counts keep every start inside the region the data support
    // (24.7 iterations, tools/debug/s1_start_perturbed.py).  TPHIP_S1_START = grid | shrunk (default) | pars | both.
    {
        bool postorder = nn <= kParsMaxNodes && desc->parent[nn - 1] < 0;
        for (int n = 0; n + 1 < nn && postorder; ++n) postorder = desc->parent[n] > n && desc->parent[n] < nn;
        const char* env = getenv("TPHIP_S1_START");
        if (postorder && !(env && !strcmp(env, "grid"))) {
            HIP_TRY(hipMemsetAsync(X.pchanges, 0, sizeof(double) * (size_t)P * (nn + 1), st));
            const int64_t maxc = p->max_locus_cols;   // (of the pattern view when one is installed)
            const unsigned ny = (unsigned)std::min<int64_t>(64, std::max<int64_t>(1, (maxc + 1023) / 1024));
            hipLaunchKernelGGL(branch_parsimony_kernel, dim3((unsigned)L, ny), dim3(128), sizeof(double) * (nn + 1), st, d_states, p->ncols,
                               p->d_offsets.p, p->d_col_weight, nn, X.parent, X.leaf, X.pchanges);
            KCHECK();
            const double alpha_env = getenv("TPHIP_S1_PARS_ALPHA") ? atof(getenv("TPHIP_S1_PARS_ALPHA")) : 0.5;
            for (int pass = 0; pass < 2; ++pass) {   // the counts shrunk towards the grid start; the counts alone (experiments)
                const bool on = pass == 0 ? !(env && !strcmp(env, "pars")) : (env && (!strcmp(env, "pars") || !strcmp(env, "both")));
                if (!on) continue;
                hipLaunchKernelGGL(grm_pars_emit_kernel, dim3((unsigned)L), dim3(64), sizeof(double) * D, st, G, X.pchanges, pass == 0 ? alpha_env : -1.0);
                KCHECK();
                RC(dr.value(G.C, G.vecs, (int64_t)L));
                hipLaunchKernelGGL(grm_pars_pick_kernel, dim3((unsigned)L), dim3(64), 0, st, G, X.counters + C_PARS, 0.0);
                KCHECK();
            }
        }
    }
    // optimiser state
    HIP_TRY(hipMemsetAsync(G.rho, 0, sizeof(double) * kHistory * P, st));
    HIP_TRY(hipMemsetAsync(G.nhist, 0, sizeof(int32_t) * P, st));
    HIP_TRY(hipMemsetAsync(G.head, 0, sizeof(int32_t) * P, st));
    HIP_TRY(hipMemsetAsync(G.kicks, 0, sizeof(int32_t) * P, st));
    HIP_TRY(hipMemsetAsync(G.iters, 0, sizeof(int32_t) * P, st));
    HIP_TRY(hipMemsetAsync(G.phase, PH_RESTART, P, st));   // RESTART: value and gradient wanted at x
    {
        std::vector<double> ones(P, 1.0), infs(P, INFINITY);
        HIP_TRY(hipMemcpyAsync(G.gamma, ones.data(), sizeof(double) * P, hipMemcpyHostToDevice, st));
        HIP_TRY(hipMemcpyAsync(G.last_df, infs.data(), sizeof(double) * P, hipMemcpyHostToDevice, st));
        HIP_TRY(hipStreamSynchronize(st));
    }
    RC(dr.zero(0, C_PARS));   // (C_PARS, the last one, was counted by the start above)
    RC(dr.set_counter(C_NLIVE, L));
    auto grm_gradient = [&](int64_t n) -> int {
        if (n <= 0) return TPHIP_OK;
        dr.nevals += n; dr.ngrads += n;
        return tphip_locus_gradient_dev(p, d_states, n, G.C.locus, G.C.exch, G.vecs, G.C.vec, G.C.scale, G.C.pidx, G.C.pfac, G.o_lnl,
                                        G.o_dexch, G.o_dlogt, G.o_sdl, G.o_d2, (void*)st);
    };
    hipLaunchKernelGGL(grm_point_kernel, dim3(L), dim3(64), 0, st, G);
    KCHECK();
    RC(grm_gradient(L));
    hipLaunchKernelGGL(grm_update_kernel, dim3(L), dim3(64), 0, st, G);
    KCHECK();
    int grm_iterations = 0;
    const int trace = getenv("TPHIP_STAGE1_TRACE") ? atoi(getenv("TPHIP_STAGE1_TRACE")) : -1;
    for (int it = 0; it < maxit_grm; ++it) {
        RC(dr.zero(C_NPEND, 3));   // NPEND, NGRAD, NFAILED
        hipLaunchKernelGGL(grm_direction_kernel, dim3(L), dim3(64), sizeof(double) * D, st, G);
        KCHECK();
        RC(dr.read_counters());
        if (it == 0 && trace >= 0) fprintf(stderr, "stage 1: the parsimony start beat the grid start on %d of %d loci\n", h_counters[C_PARS], L);
        if (h_counters[C_NLIVE] <= 0) break;
        ++grm_iterations;
        int npend = h_counters[C_NPEND];
        while (npend > 0) {
            RC(dr.zero(C_NCAND, 1));
            RC(dr.zero(C_NPEND, 1));
            hipLaunchKernelGGL(grm_trial_kernel, dim3(L), dim3(64), sizeof(double) * D, st, G);
            KCHECK();
            RC(dr.value(G.C, G.vecs, npend));
            hipLaunchKernelGGL(grm_accept_kernel, dim3(blocks_for(npend, 256)), dim3(256), 0, st, G, npend);
            KCHECK();
            RC(dr.read_counters());
            npend = h_counters[C_NPEND];
        }
        const int ngrad = h_counters[C_NGRAD];
        if (ngrad > 0) {
            RC(dr.zero(C_NCAND, 1));
            hipLaunchKernelGGL(grm_point_kernel, dim3(L), dim3(64), 0, st, G);
            KCHECK();
            RC(grm_gradient(ngrad));
        }
        hipLaunchKernelGGL(grm_update_kernel, dim3(L), dim3(64), 0, st, G);
        KCHECK();
        if (trace >= 0 && trace < L) {   // TPHIP_STAGE1_TRACE=<locus>: one line per iteration of that locus (debugging aid)
            double v[4]; int32_t w[4]; uint8_t ph;
            HIP_TRY(hipMemcpyAsync(&v[0], G.f + trace, 8, hipMemcpyDeviceToHost, st));
            HIP_TRY(hipMemcpyAsync(&v[1], G.last_df + trace, 8, hipMemcpyDeviceToHost, st));
            HIP_TRY(hipMemcpyAsync(&v[2], G.t + trace, 8, hipMemcpyDeviceToHost, st));
            HIP_TRY(hipMemcpyAsync(&v[3], G.gd + trace, 8, hipMemcpyDeviceToHost, st));
            HIP_TRY(hipMemcpyAsync(&w[0], G.nhist + trace, 4, hipMemcpyDeviceToHost, st));
            HIP_TRY(hipMemcpyAsync(&w[1], G.kicks + trace, 4, hipMemcpyDeviceToHost, st));
            HIP_TRY(hipMemcpyAsync(&w[2], G.ls_round + trace, 4, hipMemcpyDeviceToHost, st));
            HIP_TRY(hipMemcpyAsync(&w[3], G.iters + trace, 4, hipMemcpyDeviceToHost, st));
            HIP_TRY(hipMemcpyAsync(&ph, G.phase + trace, 1, hipMemcpyDeviceToHost, st));
            HIP_TRY(hipStreamSynchronize(st));
            {
                std::vector<double> hx(D), hg(D), hh(D), hdd(D);
                HIP_TRY(hipMemcpy(hx.data(), G.x + (size_t)trace * D, 8 * D, hipMemcpyDeviceToHost));
                HIP_TRY(hipMemcpy(hg.data(), G.g + (size_t)trace * D, 8 * D, hipMemcpyDeviceToHost));
                HIP_TRY(hipMemcpy(hh.data(), G.hd + (size_t)trace * D, 8 * D, hipMemcpyDeviceToHost));
                HIP_TRY(hipMemcpy(hdd.data(), G.d + (size_t)trace * D, 8 * D, hipMemcpyDeviceToHost));
                std::vector<int> idx(D);
                for (int j = 0; j < D; ++j) idx[j] = j;
                std::sort(idx.begin(), idx.end(), [&](int a, int b) { return fabs(hdd[a]) > fabs(hdd[b]); });
                fprintf(stderr, "   top |d| (of the step just taken): ");
                for (int k = 0; k < 5; ++k) fprintf(stderr, "[j %d d %.2e x %.2f g %.2e h %.2e] ", idx[k], hdd[idx[k]], hx[idx[k]], hg[idx[k]], hh[idx[k]]);
                fprintf(stderr, "\n");
            }
            fprintf(stderr, "grm it %d locus %d: f %.9f last_df %.3e t %.3e gd %.3e nhist %d kicks %d ls_rounds %d iters %d phase %d live %d\n",
                    it, trace, v[0], v[1], v[2], v[3], w[0], w[1], w[2], w[3], (int)ph, h_counters[C_NLIVE]);
        }
    }
    hipLaunchKernelGGL(grm_finish_kernel, dim3(L), dim3(64), 0, st, G, X.grm_exch, X.grm_blen, X.stash, X.grm_lnl);
    KCHECK();
    const int64_t grm_evals = dr.nevals, grm_grads = dr.ngrads;

    // ---- the 202 rate-class models --------------------------------------------------------------------------------------------
    // One quadratic model of f(rho) = lnL at the stashed lengths around the general model's optimum (31-point stencil) gives
    // every model's constrained optimum in closed form; one TRUE likelihood there decides which models can carry weight.
    hipLaunchKernelGGL(screen_emit_kernel, dim3(blocks_for((int64_t)L * kScreenPoints, 256)), dim3(256), 0, st, S);
    KCHECK();
    RC(dr.value(S.C, X.stash, (int64_t)L * kScreenPoints));
    hipLaunchKernelGGL(screen_hessian_kernel, dim3(blocks_for(L, 64)), dim3(64), 0, st, S);
    KCHECK();
    hipLaunchKernelGGL(screen_model_kernel, dim3(blocks_for((int64_t)LM, 256)), dim3(256), 0, st, S);
    KCHECK();
    RC(dr.value(S.C, X.stash, (int64_t)LM));
    hipLaunchKernelGGL(screen_score_kernel, dim3(blocks_for((int64_t)LM, 256)), dim3(256), 0, st, S);
    KCHECK();
    RC(dr.zero(0, C_COUNT));
    hipLaunchKernelGGL(screen_keep_kernel, dim3(blocks_for((int64_t)LM, 256)), dim3(256), 0, st, S, X.flags, prune);
    KCHECK();
    {
        hipcub::CountingInputIterator<int32_t> it0(0);
        size_t tmp = cub_bytes;
        HIP_TRY(hipcub::DeviceSelect::Flagged(X.cub_tmp, tmp, it0, X.flags, S.keep, X.nsel, (int)LM, st));
    }
    HIP_TRY(hipMemcpyAsync(h_counters, X.nsel, sizeof(int32_t), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    const int Q = h_counters[0];
    HIP_TRY(hipMemsetAsync(X.sub_iters, 0, sizeof(int32_t) * LM, st));
    int64_t pruned = (int64_t)LM - Q;
    int sub_iterations = 0;
    if (Q > 0) {
        B.Q = Q;
        hipLaunchKernelGGL(sub_init_kernel, dim3(blocks_for(Q, 256)), dim3(256), 0, st, B, S.theta, S.K);
        KCHECK();
        RC(dr.zero(0, C_COUNT));
        RC(dr.set_counter(C_NLIVE, Q));
        hipLaunchKernelGGL(sub_stencil_kernel, dim3(blocks_for(Q, 256)), dim3(256), 0, st, B);
        KCHECK();
        RC(dr.read_counters());
        RC(dr.value(B.C, X.stash, h_counters[C_NCAND]));
        hipLaunchKernelGGL(sub_update_kernel, dim3(blocks_for(Q, 256)), dim3(256), 0, st, B, prune);
        KCHECK();
        for (int it = 0; it < maxit_sub; ++it) {
            RC(dr.zero(C_NPEND, 3));
            hipLaunchKernelGGL(sub_direction_kernel, dim3(blocks_for(Q, 256)), dim3(256), 0, st, B);
            KCHECK();
            RC(dr.read_counters());
            if (h_counters[C_NLIVE] <= 0) break;
            ++sub_iterations;
            int npend = h_counters[C_NPEND];
            while (npend > 0) {
                RC(dr.zero(C_NCAND, 1));
                RC(dr.zero(C_NPEND, 1));
                hipLaunchKernelGGL(sub_trial_kernel, dim3(blocks_for(Q, 256)), dim3(256), 0, st, B);
                KCHECK();
                RC(dr.value(B.C, X.stash, npend));
                hipLaunchKernelGGL(sub_accept_kernel, dim3(blocks_for(npend, 256)), dim3(256), 0, st, B, npend);
                KCHECK();
                RC(dr.read_counters());
                npend = h_counters[C_NPEND];
            }
            if (h_counters[C_NGRAD] > 0) {
                RC(dr.zero(C_NCAND, 1));
                hipLaunchKernelGGL(sub_stencil_kernel, dim3(blocks_for(Q, 256)), dim3(256), 0, st, B);
                KCHECK();
                RC(dr.read_counters());
                RC(dr.value(B.C, X.stash, h_counters[C_NCAND]));
            }
            hipLaunchKernelGGL(sub_update_kernel, dim3(blocks_for(Q, 256)), dim3(256), 0, st, B, prune);
            KCHECK();
            if (prune) {
                hipLaunchKernelGGL(sub_prune_kernel, dim3(blocks_for(Q, 256)), dim3(256), 0, st, B);
                KCHECK();
            }
        }
        RC(dr.read_counters());
        pruned += h_counters[C_NPRUNED];
        hipLaunchKernelGGL(sub_scatter_kernel, dim3(blocks_for(Q, 256)), dim3(256), 0, st, B, S.theta, S.f_at, X.sub_iters);
        KCHECK();
    }
    // ---- Akaike weights, averaged rates (bf:806-847) ----------------------------------------------------------------------------
    hipLaunchKernelGGL(average_kernel, dim3(blocks_for(L, 64)), dim3(64), 0, st, L, X.grm_exch, X.grm_lnl, S.theta, S.f_at, X.cls, X.kk,
                       X.o_exch, weights_out ? X.o_weights : nullptr, lnl_out ? X.o_lnl : nullptr, model_exch_out ? X.o_mexch : nullptr);
    KCHECK();
    HIP_TRY(hipMemcpyAsync(exch_out, X.o_exch, sizeof(double) * P * 6, hipMemcpyDeviceToHost, st));
    if (pi_out) {
        if (opt.empirical_pi) HIP_TRY(hipMemcpyAsync(pi_out, X.pi, sizeof(double) * P * 4, hipMemcpyDeviceToHost, st));
        else memcpy(pi_out, desc->pi, sizeof(double) * P * 4);
    }
    if (weights_out) HIP_TRY(hipMemcpyAsync(weights_out, X.o_weights, sizeof(double) * P * kModels, hipMemcpyDeviceToHost, st));
    if (lnl_out) HIP_TRY(hipMemcpyAsync(lnl_out, X.o_lnl, sizeof(double) * P * kModels, hipMemcpyDeviceToHost, st));
    if (model_exch_out) HIP_TRY(hipMemcpyAsync(model_exch_out, X.o_mexch, sizeof(double) * P * kModels * 6, hipMemcpyDeviceToHost, st));
    if (grm_blen_out) HIP_TRY(hipMemcpyAsync(grm_blen_out, X.grm_blen, sizeof(double) * P * nn, hipMemcpyDeviceToHost, st));
    if (grm_iters_out) HIP_TRY(hipMemcpyAsync(grm_iters_out, G.iters, sizeof(int32_t) * P, hipMemcpyDeviceToHost, st));
    if (sub_iters_out) HIP_TRY(hipMemcpyAsync(sub_iters_out, X.sub_iters, sizeof(int32_t) * LM, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    if (stats_out) {
        stats_out[0] = dr.nevals; stats_out[1] = dr.ngrads; stats_out[2] = grm_evals; stats_out[3] = grm_grads;
        stats_out[4] = pruned; stats_out[5] = Q; stats_out[6] = grm_iterations; stats_out[7] = sub_iterations;
    }
    return TPHIP_OK;
}

extern "C" int tphip_stage1_fit(tphip_plan* p, const uint8_t* states, void** d_states_cache, const tphip_stage1_opts* opts,
                                double* exch_out, double* pi_out, double* weights_out, double* lnl_out, double* model_exch_out,
                                double* grm_blen_out, int32_t* grm_iters_out, int32_t* sub_iters_out, int64_t* stats_out) {
    if (!p || !states) return fail(TPHIP_ERR_INVALID, "null argument");
    HIP_TRY(hipSetDevice(p->device));
    tphip_stage1_opts opt;
    memset(&opt, 0, sizeof(opt));
    if (opts) {
        if (opts->struct_size < sizeof(uint32_t) || opts->struct_size > 4096)
            return fail(TPHIP_ERR_INVALID, "tphip_stage1_opts.struct_size not set");
        memcpy(&opt, opts, std::min<size_t>(opts->struct_size, sizeof(opt)));
    }
    const size_t n = (size_t)p->ncols, pitch = opt.row_pitch > 0 ? (size_t)opt.row_pitch : n;
    if (pitch < n) return fail(TPHIP_ERR_INVALID, "row_pitch smaller than the plan's column count");
    if (pitch != n || opt.compress_patterns) {
        // nothing of this upload is worth keeping: a column range of a bigger array, or columns that are collapsed into
        // patterns right away
        if (d_states_cache) return fail(TPHIP_ERR_INVALID, "d_states_cache must be NULL with row_pitch / compress_patterns");
        uint8_t* d_s = nullptr;
        HIP_TRY(hipMalloc((void**)&d_s, n * (size_t)p->ntaxa + 1));
        hipError_t e = pitch == n ? hipMemcpy(d_s, states, n * (size_t)p->ntaxa, hipMemcpyHostToDevice)
                                  : hipMemcpy2D(d_s, n, states, pitch, n, (size_t)p->ntaxa, hipMemcpyHostToDevice);
        int rc = e == hipSuccess ? TPHIP_OK : fail(TPHIP_ERR_HIP, std::string("upload of the alignment: ") + hipGetErrorString(e));
        if (!rc)
            rc = tphip_stage1_fit_dev(p, d_s, &opt, exch_out, pi_out, weights_out, lnl_out, model_exch_out, grm_blen_out, grm_iters_out,
                                      sub_iters_out, stats_out, nullptr);
        (void)hipFree(d_s);
        return rc;
    }
    void* local = nullptr;
    void** cache = d_states_cache ? d_states_cache : &local;
    uint8_t* d_s = nullptr;
    int rc = tphip_internal_stage_alignment(p, states, cache, &d_s);
    if (!rc)
        rc = tphip_stage1_fit_dev(p, d_s, &opt, exch_out, pi_out, weights_out, lnl_out, model_exch_out, grm_blen_out, grm_iters_out,
                                  sub_iters_out, stats_out, nullptr);
    if (!d_states_cache && local) {
        const int rf = tphip_free_device(p, local);
        if (!rc) rc = rf;
    }
    return rc;
}
