// tphip.hip -- C ABI (include/tphip.h) over the gfx950 kernels.  Host side only does validation, the
// one-off tree compilation, table uploads and kernel launches; there is no CPU compute path.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstddef>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "gtr_model.hpp"
#include "gtr_setup_kernel.hpp"
#include "locus_lik_params.hpp"
#include "locus_value_params.hpp"
#include "pattern_kernels.hpp"
#include <hipcub/hipcub.hpp>
#include "pi_kernels.hpp"
#include "site_rate_params.hpp"
#include "tphip.h"
#include "tree_program.hpp"
#include "tphip_internal.hpp"

using namespace tphip;

thread_local std::string g_tphip_err;

static void free_host_buffers(struct tphip_plan* p);
static void free_parts(struct tphip_plan* p);

extern "C" {

int tphip_version(void) { return TPHIP_VERSION; }

const char* tphip_last_error(void) { return g_tphip_err.c_str(); }

int tphip_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int tphip_plan_destroy(tphip_plan* plan) {
    if (!plan) return TPHIP_OK;
    (void)hipSetDevice(plan->device);
    plan->d_ops.release(); plan->d_fused_ops.release(); plan->d_models.release(); plan->d_offsets.release();
    plan->d_locus_pichunk_offsets.release(); plan->d_site_chunk_locus.release(); plan->d_site_chunk_index.release();
    plan->d_pi_chunk_locus.release(); plan->d_pi_chunk_index.release(); plan->d_times.release();
    plan->d_intervals.release(); plan->d_evals.release(); plan->d_tip_taxon.release(); plan->d_op_node.release();
    plan->d_lik_ops.release();
    plan->d_value_ops.release();
    plan->d_value_tip_node.release();
    plan->d_cat.release();
    free_host_buffers(plan);
    free_parts(plan);
    if (plan->d_tape) { (void)hipFree(plan->d_tape); plan->d_tape = nullptr; }
    if (plan->d_value_ws) { (void)hipFree(plan->d_value_ws); plan->d_value_ws = nullptr; }
    if (plan->d_grad_eig) { (void)hipFree(plan->d_grad_eig); plan->d_grad_eig = nullptr; }
    if (plan->d_value_packed) { (void)hipFree(plan->d_value_packed); plan->d_value_packed = nullptr; }
    if (plan->d_part) { (void)hipFree(plan->d_part); plan->d_part = nullptr; }
    if (plan->d_col_weight) { (void)hipFree(plan->d_col_weight); plan->d_col_weight = nullptr; }
    if (plan->d_grad_params) { (void)hipFree(plan->d_grad_params); plan->d_grad_params = nullptr; }
    plan->d_grad2_fops.release(); plan->d_grad2_rops.release();
    if (plan->d_grad2_ws) { (void)hipFree(plan->d_grad2_ws); plan->d_grad2_ws = nullptr; }
    if (plan->d_grad2_params) { (void)hipFree(plan->d_grad2_params); plan->d_grad2_params = nullptr; }
    if (plan->d_arena) { (void)hipFree(plan->d_arena); plan->d_arena = nullptr; }
    if (plan->h_arena) { (void)hipHostFree(plan->h_arena); plan->h_arena = nullptr; }
    for (hipEvent_t e : plan->ev) (void)hipEventDestroy(e);
    delete plan;
    return TPHIP_OK;
}

struct tphip_saved_desc {
    tphip_plan_desc d;
    std::vector<int32_t> parent, leaf, times, intervals;
    std::vector<double> blen, pi, exch, cat_rate, cat_weight;
    std::vector<int64_t> offsets;
};

static tphip_saved_desc* save_desc(const tphip_plan_desc* d) {
    tphip_saved_desc* s = new tphip_saved_desc();
    s->d = *d;
    s->parent.assign(d->parent, d->parent + d->nnodes);
    s->blen.assign(d->branch_len, d->branch_len + d->nnodes);
    s->leaf.assign(d->leaf_taxon, d->leaf_taxon + d->nnodes);
    s->offsets.assign(d->locus_offsets, d->locus_offsets + d->nloci + 1);
    s->pi.assign(d->pi, d->pi + 4 * d->nloci);
    s->exch.assign(d->exch, d->exch + 6 * d->nloci);
    if (d->n_t) s->times.assign(d->times, d->times + d->n_t);
    if (d->n_i) s->intervals.assign(d->intervals, d->intervals + 2 * (size_t)d->n_i);
    if (d->ncat > 0 && d->cat_rate && d->cat_weight) {
        s->cat_rate.assign(d->cat_rate, d->cat_rate + d->ncat);
        s->cat_weight.assign(d->cat_weight, d->cat_weight + d->ncat);
    }
    // the saved descriptor points at its own copies (the caller's arrays may be gone after tphip_plan_create returns)
    s->d.parent = s->parent.data(); s->d.branch_len = s->blen.data(); s->d.leaf_taxon = s->leaf.data();
    s->d.locus_offsets = s->offsets.data(); s->d.pi = s->pi.data(); s->d.exch = s->exch.data();
    s->d.times = s->times.empty() ? nullptr : s->times.data();
    s->d.intervals = s->intervals.empty() ? nullptr : s->intervals.data();
    s->d.cat_rate = s->cat_rate.empty() ? nullptr : s->cat_rate.data();
    s->d.cat_weight = s->cat_weight.empty() ? nullptr : s->cat_weight.data();
    return s;
}

// for the other translation units of the library (stage1_driver.hip)
const tphip_plan_desc* tphip_internal_saved_desc(const tphip_plan* p) { return (p && p->saved) ? &p->saved->d : nullptr; }

static void free_parts(tphip_plan* p) {
    for (tphip_plan* q : p->parts) (void)tphip_plan_destroy(q);
    p->parts.clear();
    delete p->saved;
    p->saved = nullptr;
}

// The plans of the locus groups of a host-pointer run (created on first use): boundaries at the loci nearest to equal
// column counts; each group is an ordinary plan over its own loci (same tree, schedule and options).
static int make_parts(tphip_plan* p) {
    if (!p->parts.empty()) return TPHIP_OK;
    const tphip_saved_desc& S = *p->saved;
    const int64_t L = p->nloci, n = p->ncols;
    const int K = (int)std::min<int64_t>(p->host_split, L);
    std::vector<int64_t> cut(1, 0);
    for (int k = 1; k < K; ++k) {
        const int64_t target = n * k / K;
        int64_t l = std::lower_bound(S.offsets.begin(), S.offsets.end(), target) - S.offsets.begin();
        l = std::max<int64_t>(cut.back() + 1, std::min<int64_t>(l, L - (K - k)));
        cut.push_back(l);
    }
    cut.push_back(L);
    for (size_t k = 0; k + 1 < cut.size(); ++k)
        if (S.offsets[cut[k + 1]] == S.offsets[cut[k]]) { p->host_split = 1; return TPHIP_OK; }   // a group without columns: no pipeline
    for (size_t k = 0; k + 1 < cut.size(); ++k) {
        const int64_t l0 = cut[k], l1 = cut[k + 1];
        std::vector<int64_t> off(S.offsets.begin() + l0, S.offsets.begin() + l1 + 1);
        for (int64_t& o : off) o -= S.offsets[l0];
        tphip_plan_desc d = S.d;
        d.parent = S.parent.data(); d.branch_len = S.blen.data(); d.leaf_taxon = S.leaf.data();
        d.nloci = l1 - l0; d.locus_offsets = off.data(); d.pi = S.pi.data() + 4 * l0; d.exch = S.exch.data() + 6 * l0;
        d.times = S.times.empty() ? nullptr : S.times.data();
        d.intervals = S.intervals.empty() ? nullptr : S.intervals.data();
        d.cat_rate = S.cat_rate.empty() ? nullptr : S.cat_rate.data();
        d.cat_weight = S.cat_weight.empty() ? nullptr : S.cat_weight.data();
        tphip_plan* q = nullptr;
        const int rc = tphip_plan_create(&d, &q);
        if (rc) {   // whatever stops a group's plan: run the batch whole (the plain path reports real errors itself)
            for (tphip_plan* r : p->parts) (void)tphip_plan_destroy(r);
            p->parts.clear();
            p->host_split = 1;
            return TPHIP_OK;
        }
        q->is_part = true;
        p->parts.push_back(q);
    }
    p->part_locus = cut;
    return TPHIP_OK;
}

int tphip_plan_create(const tphip_plan_desc* d_in, tphip_plan** out) {
    if (!d_in || !out) return fail(TPHIP_ERR_INVALID, "null plan descriptor");
    *out = nullptr;
    // the caller's struct may be shorter (built against an older header): its missing trailing fields are zero
    if (d_in->struct_size < offsetof(tphip_plan_desc, round_decimals) + sizeof(int32_t) || d_in->struct_size > 4096)
        return fail(TPHIP_ERR_INVALID, "tphip_plan_desc.struct_size not set (zero-initialise the struct, then set it to sizeof)");
    tphip_plan_desc d_copy;
    memset(&d_copy, 0, sizeof(d_copy));
    memcpy(&d_copy, d_in, std::min<size_t>(d_in->struct_size, sizeof(d_copy)));
    d_copy.struct_size = (uint32_t)sizeof(d_copy);
    const tphip_plan_desc* d = &d_copy;
    int ndev = tphip_device_count();
    if (ndev <= 0) return fail(TPHIP_ERR_NO_DEVICE, "no HIP device visible: libtphip has no CPU path");
    if (d->device < 0 || d->device >= ndev) return fail(TPHIP_ERR_INVALID, "device ordinal out of range");
    if (d->ntaxa < 2) return fail(TPHIP_ERR_INVALID, "ntaxa must be >= 2");
    if (!d->parent || !d->branch_len || !d->leaf_taxon) return fail(TPHIP_ERR_INVALID, "null tree arrays");
    if (d->nloci < 1 || !d->locus_offsets || !d->pi || !d->exch) return fail(TPHIP_ERR_INVALID, "null or empty locus arrays");
    if (d->T < 0 || d->n_t < 0 || d->n_i < 0) return fail(TPHIP_ERR_INVALID, "negative schedule size");
    if ((d->n_t && !d->times) || (d->n_i && !d->intervals)) return fail(TPHIP_ERR_INVALID, "null times/intervals");
    if (!(d->correction > 0.0)) return fail(TPHIP_ERR_INVALID, "correction must be > 0");
    if (d->integ_mode != TPHIP_INTEG_QUADPACK && d->integ_mode != TPHIP_INTEG_CLOSED)
        return fail(TPHIP_ERR_INVALID, "unknown integ_mode");
    if (d->start_rule != TPHIP_START_AUTO && d->start_rule != TPHIP_START_PARSIMONY && d->start_rule != TPHIP_START_REFERENCE)
        return fail(TPHIP_ERR_INVALID, "unknown start_rule");
    if (d->pattern_dedup != TPHIP_DEDUP_AUTO && d->pattern_dedup != TPHIP_DEDUP_OFF && d->pattern_dedup != TPHIP_DEDUP_ON)
        return fail(TPHIP_ERR_INVALID, "unknown pattern_dedup");
    for (int i = 0; i < d->n_t; ++i)
        if (d->times[i] < 0 || d->times[i] >= d->T)  // numpy would raise IndexError (tapir/compute.py:78)
            return fail(TPHIP_ERR_INVALID, "a --times value is outside 0..T-1 (index out of bounds for the net PI vector)");
    for (int i = 0; i < d->n_i; ++i)
        if (!(d->intervals[2 * i] < d->intervals[2 * i + 1]))  // assert at tapir/compute.py:90-91
            return fail(TPHIP_ERR_INVALID, "Start time is sooner than end time in an interval");
    if (d->locus_offsets[0] != 0) return fail(TPHIP_ERR_INVALID, "locus_offsets[0] must be 0");
    for (int64_t l = 0; l < d->nloci; ++l) {
        if (d->locus_offsets[l + 1] < d->locus_offsets[l]) return fail(TPHIP_ERR_INVALID, "locus_offsets not monotone");
        int npos = 0;
        for (int k = 0; k < 4; ++k) {
            if (!(d->pi[l * 4 + k] >= 0.0) || !std::isfinite(d->pi[l * 4 + k]))
                return fail(TPHIP_ERR_INVALID, "base frequencies of locus " + std::to_string(l) + " must be finite and >= 0");
            npos += d->pi[l * 4 + k] > 0.0;
        }
        if (npos < 2) return fail(TPHIP_ERR_INVALID, "locus " + std::to_string(l) + " has fewer than two bases with a positive frequency");
        for (int k = 0; k < 6; ++k)
            if (!(d->exch[l * 6 + k] >= 0.0)) return fail(TPHIP_ERR_INVALID, "exchangeabilities must be >= 0");
    }
    const int64_t ncols = d->locus_offsets[d->nloci];
    if (ncols >= (int64_t)1 << 31) return fail(TPHIP_ERR_INVALID, "more than 2^31-1 columns in one batch");
    std::vector<double> cat;
    if (d->ncat > 1) {
        if (d->ncat > 16) return fail(TPHIP_ERR_INVALID, "at most 16 rate categories");
        if (!d->cat_rate || !d->cat_weight) return fail(TPHIP_ERR_INVALID, "rate categories need cat_rate and cat_weight");
        double wsum = 0;
        for (int k = 0; k < d->ncat; ++k) {
            if (!(d->cat_rate[k] > 0.0) || !(d->cat_weight[k] > 0.0) || !std::isfinite(d->cat_rate[k]) || !std::isfinite(d->cat_weight[k]))
                return fail(TPHIP_ERR_INVALID, "category rates and weights must be positive and finite");
            wsum += d->cat_weight[k];
        }
        cat.resize(2 * (size_t)d->ncat);
        for (int k = 0; k < d->ncat; ++k) { cat[k] = d->cat_rate[k]; cat[d->ncat + k] = std::log(d->cat_weight[k] / wsum); }
    }

    tphip_plan* p = new tphip_plan();
    p->device = d->device; p->ntaxa = d->ntaxa; p->nloci = d->nloci; p->ncols = ncols;
    p->T = d->T; p->n_t = d->n_t; p->n_i = d->n_i; p->integ_mode = d->integ_mode;
    p->threshold = d->threshold; p->round_decimals = d->round_decimals; p->correction = d->correction;
    // TPHIP_START_AUTO: HyPhy's own start where the parsimony start was ever seen to end on another local optimum (small trees)
    // (the rate mixture is an extension without a HyPhy counterpart to start like: it keeps the parsimony start)
    p->start_rule = d->start_rule == TPHIP_START_AUTO
                        ? ((d->ntaxa >= kFirstStepMinTaxa || d->ncat > 1) ? TPHIP_START_PARSIMONY : TPHIP_START_REFERENCE)
                        : d->start_rule;
    p->dedup_mode = d->pattern_dedup;
    p->ncat = cat.empty() ? 0 : d->ncat;
    std::string terr = build_tree_program(d->ntaxa, d->nnodes, d->parent, d->branch_len, d->leaf_taxon, &p->prog);
    if (!terr.empty()) { delete p; return fail(TPHIP_ERR_INVALID, "tree: " + terr); }
    if (p->prog.nleaves != d->ntaxa) {  // HyPhy refuses a tree / alignment mismatch as well
        delete p;
        return fail(TPHIP_ERR_INVALID, "tree: the number of leaves differs from the number of alignment rows");
    }
    const size_t lds_bytes = (kSiteLdsHeader + (size_t)p->prog.stack_depth * 12 * kSiteBlock) * sizeof(double);
    if (lds_bytes > 160 * 1024) { delete p; return fail(TPHIP_ERR_INVALID, "tree needs a deeper LDS stack than 160 KiB allows"); }
    p->h_offsets.assign(d->locus_offsets, d->locus_offsets + d->nloci + 1);
    for (int64_t l = 0; l < d->nloci; ++l)
        p->max_locus_cols = std::max<int64_t>(p->max_locus_cols, d->locus_offsets[l + 1] - d->locus_offsets[l]);

    // chunk tables: work slices for the optimiser, 1024-column chunks for classification and PI.
    // A slice is what one wave works through with lane refill: long enough to amortise the drain at its end,
    // short enough that the batch still makes >= ~8192 waves (32 per CU).
    {
        // Target slice length of the non-persistent mode (small batches) and of the eval diagnostic.
        int64_t cc = 512;
        if (const char* e = getenv("TPHIP_SITE_CHUNK")) {  // tuning knob for experiments
            long v = atol(e);
            if (v >= kSiteBlock) cc = v;
        }
        p->site_chunk_cols = (int32_t)cc;
    }
    std::vector<int32_t> scl, sci, pcl, pci;
    std::vector<int64_t> lpo(d->nloci + 1, 0);
    for (int64_t l = 0; l < d->nloci; ++l) {
        const int64_t S = p->h_offsets[l + 1] - p->h_offsets[l];
        const int64_t ns_max = std::max<int64_t>(1, (S + p->site_chunk_cols / 2) / p->site_chunk_cols);
        for (int64_t c = 0; c < ns_max && S > 0; ++c) { scl.push_back((int32_t)l); sci.push_back((int32_t)c); }
        for (int64_t c = 0; c * kPiChunk < S; ++c) { pcl.push_back((int32_t)l); pci.push_back((int32_t)c); }
        lpo[l + 1] = (int64_t)pcl.size();
    }
    p->n_site_chunks = (int64_t)scl.size();
    p->n_pi_chunks = (int64_t)pcl.size();

    hipError_t e = hipSetDevice(d->device);
    std::vector<int32_t> times(d->times, d->times + d->n_t), iv(d->intervals, d->intervals + 2 * (size_t)d->n_i);
    DevBuf<double> d_pi, d_exch;
    std::vector<double> hpi(d->pi, d->pi + 4 * (size_t)d->nloci), hex(d->exch, d->exch + 6 * (size_t)d->nloci);
    // A base that never occurs in a (short) locus has empirical frequency 0 (HarvestFrequencies, bf:968); HyPhy takes
    // that as it is: the state is unreachable and no tip carries it, so the likelihood is that of the three-state
    // model.  The eigen-form used here needs D^-1/2, so the zero is floored at kPiFloor before the renormalisation in
    // gtr_setup_kernel: the dead state then enters every likelihood with weight O(1e-12), below the 1e-9 the parity
    // tests resolve.
    for (int64_t l = 0; l < d->nloci; ++l) {
        double sum = 0;
        for (int k = 0; k < 4; ++k) sum += hpi[4 * l + k];
        for (int k = 0; k < 4; ++k) hpi[4 * l + k] = std::max(hpi[4 * l + k], kPiFloor * sum);
    }
    std::vector<int32_t> tip_taxon;
    for (const TreeOp& op : p->prog.ops) if (op.code <= OP_TIP_MUL) tip_taxon.push_back(op.taxon);
    p->nwords = (int32_t)((tip_taxon.size() + 7) / 8);
    while (tip_taxon.size() % 8) tip_taxon.push_back(tip_taxon.back());   // classify_kernel reads it eight at a time
    if (e == hipSuccess) e = p->d_tip_taxon.upload(tip_taxon);
    if (e == hipSuccess) e = p->d_op_node.upload(p->prog.op_node);
    if (e == hipSuccess && !cat.empty()) e = p->d_cat.upload(cat);
    if (e == hipSuccess) {
        std::vector<int4> lops(p->prog.ops.size());
        for (size_t i = 0; i < lops.size(); ++i) {
            const int32_t code = p->prog.ops[i].code;
            lops[i] = make_int4(code, p->prog.ops[i].taxon, p->prog.op_node[i],
                                code == OP_POP_MUL ? p->prog.op_partner[i] : p->prog.op_tape[i]);
        }
        e = p->d_lik_ops.upload(lops);
    }
    if (e == hipSuccess) {
        // locus_value_kernel's stream (build_value_program, locus_value_params.hpp)
        const auto& ops = p->prog.ops;
        std::vector<int4> vops;
        std::vector<int32_t> tip_node;
        const bool ok = build_value_program(p->prog, d->ntaxa, d->nnodes, &vops, &tip_node);
        if (ok) e = p->d_value_tip_node.upload(tip_node);
        p->value_nops = ok ? (int32_t)vops.size() : 0;
        if (ok && e == hipSuccess) e = p->d_value_ops.upload(vops);
        // the transition-matrix gradient kernel's programs: the same forward stream with a tape slot on every BRANCH, and the
        // pre-order walk over the internal nodes for the reverse sweep (binary trees only)
        if (ok && e == hipSuccess) {
            std::vector<int4> fops(vops);
            std::vector<int32_t> tape_slot(d->nnodes, -1), tip_pos(d->ntaxa, -1);
            int32_t nslot = 0, pos = 0;
            for (int4& o : fops)
                if ((o.x & OP_CODE_MASK) == OP_BRANCH) { tape_slot[o.y / 128] = nslot; o.z = nslot++; }
            for (size_t i = 0; i < ops.size(); ++i)
                if (ops[i].code <= OP_TIP_MUL) tip_pos[ops[i].taxon] = pos++;
            std::vector<int4> rops;
            int32_t rdepth = 0;
            const std::string gerr = build_grad2_program(d->nnodes, d->parent, d->leaf_taxon, tape_slot, tip_pos, &rops, &rdepth);
            if (gerr.empty() && rdepth <= kGrad2MaxRDepth && !getenv("TPHIP_GRAD_EIGENBASIS")) {
                e = p->d_grad2_fops.upload(fops);
                if (e == hipSuccess) e = p->d_grad2_rops.upload(rops);
                p->grad2_nrops = (int32_t)(rops.size() / 2);
                p->grad2_ntape = nslot;
                p->grad2_rdepth = rdepth;
                p->grad2_ok = e == hipSuccess;
            }
        }
    }
    p->nnodes = d->nnodes;
    if (e == hipSuccess) e = p->d_ops.upload(p->prog.ops);
    if (e == hipSuccess) e = p->d_fused_ops.upload(p->prog.fused_ops);
    if (e == hipSuccess) e = p->d_offsets.upload(p->h_offsets);
    if (e == hipSuccess) e = p->d_locus_pichunk_offsets.upload(lpo);
    if (e == hipSuccess) e = p->d_site_chunk_locus.upload(scl);
    if (e == hipSuccess) e = p->d_site_chunk_index.upload(sci);
    if (e == hipSuccess) e = p->d_pi_chunk_locus.upload(pcl);
    if (e == hipSuccess) e = p->d_pi_chunk_index.upload(pci);
    if (e == hipSuccess) e = p->d_times.upload(times);
    if (e == hipSuccess) e = p->d_intervals.upload(iv);
    if (e == hipSuccess) e = p->d_models.alloc((size_t)d->nloci);
    if (e == hipSuccess) e = p->d_evals.alloc(8);   // [0] evaluations; [1..5] diagnostics of a TPHIP_SITE_TRACE_ROUNDS build
    if (e == hipSuccess) e = hipMemset(p->d_evals.p, 0, 8 * sizeof(unsigned long long));
    if (e == hipSuccess) e = d_pi.upload(hpi);
    if (e == hipSuccess) e = d_exch.upload(hex);
    if (e == hipSuccess) {
        const int bs = 128;
        gtr_setup_kernel<<<dim3((unsigned)((d->nloci + bs - 1) / bs)), dim3(bs)>>>(d_pi.p, d_exch.p, d->nloci, p->d_models.p);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipDeviceSynchronize();
    d_pi.release();
    d_exch.release();
    if (e != hipSuccess) {
        std::string m = std::string("plan setup: ") + hipGetErrorString(e);
        tphip_plan_destroy(p);
        return fail(TPHIP_ERR_HIP, m);
    }
    {   // persistent grid: exactly the waves the device keeps resident for this LDS footprint
        hipDeviceProp_t prop;
        int per_cu = 0;
        if (hipGetDeviceProperties(&prop, d->device) != hipSuccess) { tphip_plan_destroy(p); return fail(TPHIP_ERR_HIP, "hipGetDeviceProperties failed"); }
        const hipError_t oe = site_rate_kernel_occupancy(p->nwords <= 2 ? 2 : p->nwords <= 8 ? 8 : kStreamWords, lds_bytes, &per_cu);
        if (oe != hipSuccess || per_cu < 1) per_cu = 1;
        p->site_waves = per_cu * prop.multiProcessorCount;
        p->num_cus = prop.multiProcessorCount;
        // Deep trees (256 taxa: 4 parked partials = 24 KB per wave) are capped at 6 waves per CU by the LDS stack, not by
        // the registers.  The deepest slots are the least used (C5 tree: 9 of 80 pushes per evaluation reach the fourth),
        // so the streamed-words kernel has a variant that parks them in a global scratch row of the wave instead: with
        // three slots in LDS (19.7 KB) eight waves fit.  Only worth it for the persistent grid (one scratch row per wave).
        p->site_lds_depth = p->prog.stack_depth;
        {
            const int keep = (int)((160 * 1024 / 8 - kSiteLdsHeader * sizeof(double)) / (12 * kSiteBlock * sizeof(double)));   // = 3
            bool want = p->nwords > 8 && p->prog.stack_depth > keep && per_cu < 8;
            if (const char* es = getenv("TPHIP_SITE_SPILL")) want = (es[0] == '1') && p->nwords > 8 && p->prog.stack_depth > 1;
            if (want) {
                const int depth = std::max(1, std::min(keep, p->prog.stack_depth - 1));
                const size_t lds2 = (kSiteLdsHeader + (size_t)depth * 12 * kSiteBlock) * sizeof(double);
                int per2 = 0;
                if (site_rate_kernel_occupancy(kStreamWordsSpill, lds2, &per2) == hipSuccess && per2 > per_cu &&
                    ncols / ((int64_t)per2 * prop.multiProcessorCount) >= 1000) {   // persistent mode will be chosen below
                    p->site_lds_depth = depth;
                    p->site_waves = per2 * prop.multiProcessorCount;
                }
            }
        }
        // locus likelihood (value) kernel: staging the state masks in LDS costs it more than it saves (measured 13.6 ms
        // vs 10.0 ms per 1616 candidates x 20000 columns x 64 taxa); opt-in for experiments
        p->lik_lds = ((size_t)p->nnodes * 4 + (size_t)p->prog.stack_depth * 4 * kLikBlock) * sizeof(double);
        const size_t lik_stage_bytes = (size_t)p->ntaxa * kLikBlock;
        p->lik_stage = (getenv("TPHIP_LIK_STAGE") && lik_stage_bytes <= 48 * 1024 && p->lik_lds + lik_stage_bytes <= 150 * 1024) ? 1 : 0;
        if (p->lik_stage) p->lik_lds += lik_stage_bytes;
        p->lik_ok = p->lik_lds <= 150 * 1024;
        if (p->lik_ok && p->lik_lds > 64 * 1024)
            p->lik_ok = locus_loglik_kernel_allow_lds(150 * 1024) == hipSuccess;
        // value kernel on transition matrices (locus_value_kernel.hpp): C columns per thread share the op decode and the
        // scalar loads of a branch's matrix; parked siblings live in registers (template depth), LDS holds the tip matrices
        // and the packed state masks
        {
            int cols = p->max_locus_cols >= 512 ? 2 : 1;
            if (const char* env = getenv("TPHIP_VALUE_COLS")) cols = atoi(env) >= 2 ? 2 : 1;
            p->value_cols = 0;
            if (p->value_nops > 0 && p->prog.stack_depth <= kValueMaxDepth && !getenv("TPHIP_VALUE_EIGENBASIS")) {
                const size_t need = (size_t)p->ntaxa * kValueTipRow * sizeof(double) + (size_t)p->nwords * cols * kLikBlock * sizeof(uint32_t);
                if (need <= 96 * 1024 &&
                    (need <= 64 * 1024 || locus_value_kernel_allow_lds(cols, p->prog.stack_depth, 96 * 1024) == hipSuccess)) {
                    p->value_cols = cols;
                    p->value_lds = need;
                }
            }
        }
        // gradient kernel: staging helps it (95.6 -> 91.7 ms)
        p->grad_slots = kGradSlots;   // fewer accumulator addresses per branch when the tree would not fit otherwise
        while (p->grad_slots > 1 && (size_t)p->nnodes * (kGradEF + 2 * kGradWaves * p->grad_slots) * sizeof(double) > 100 * 1024)
            p->grad_slots >>= 1;
        p->grad_lds = (size_t)p->nnodes * (kGradEF + 2 * kGradWaves * p->grad_slots) * sizeof(double);
        const size_t grad_stage_bytes = (size_t)p->ntaxa * kGradBlock;
        p->grad_stage = (!getenv("TPHIP_LIK_NO_STAGE") && grad_stage_bytes <= 48 * 1024 && p->grad_lds + grad_stage_bytes <= 150 * 1024) ? 1 : 0;
        if (p->grad_stage) p->grad_lds += grad_stage_bytes;
        p->grad_ok = p->grad_lds <= 150 * 1024;
        if (p->grad_ok && p->grad_lds > 64 * 1024)
            p->grad_ok = locus_grad_kernel_allow_lds(150 * 1024) == hipSuccess;
        if (p->grad_ok) {
            int bpc = 1;
            if (locus_grad_kernel_occupancy(p->grad_lds, &bpc) != hipSuccess || bpc < 1) bpc = 1;
            // the tape of the resident blocks should stay within reach of the 256 MB memory-side cache: with 64 taxa, 4
            // blocks per CU (390 MB of tape) ran slower than 3 (91 vs 85 ms); fewer than 3 loses more to latency
            const size_t tape_per_block = (size_t)std::max(1, p->prog.ntape + p->prog.stack_depth) * 4 * kGradBlock * sizeof(double);
            if (tape_per_block * (size_t)p->num_cus * (size_t)bpc > ((size_t)300 << 20)) bpc = std::min(bpc, 3);
            if (const char* env = getenv("TPHIP_GRAD_BLOCKS_PER_CU")) bpc = std::max(1, atoi(env));
            p->grad_blocks_per_cu = bpc;
        }
        // transition-matrix gradient kernel: LDS = tip table + tipY + parked adjoints + per-branch accumulators + state codes
        if (p->grad2_ok) {
            p->grad2_lds = ((size_t)p->ntaxa * kValueTipRow + 64 + (size_t)p->grad2_rdepth * 4 * kGrad2Block +
                            2 * (size_t)kGrad2Waves * p->nnodes) * sizeof(double) + (size_t)p->nwords * kGrad2Block * sizeof(uint32_t);
            p->grad2_ok = p->value_cols > 0 && p->grad2_lds <= 96 * 1024 &&
                          (p->grad2_lds <= 64 * 1024 || locus_grad2_kernel_allow_lds(p->prog.stack_depth, 96 * 1024) == hipSuccess);
            if (p->grad2_ok) {
                int bpc = 1;
                if (locus_grad2_kernel_occupancy(p->prog.stack_depth, p->grad2_lds, &bpc) != hipSuccess || bpc < 1) bpc = 1;
                if (const char* env = getenv("TPHIP_GRAD2_BLOCKS_PER_CU")) bpc = std::max(1, atoi(env));
                p->grad2_blocks_per_cu = bpc;
            }
        }
        if (const char* env = getenv("TPHIP_LIK_NSPLIT")) p->lik_nsplit_forced = std::max(1, atoi(env));
        if (const char* fb = getenv("TPHIP_FORCE_BYTE_PATH")) p->force_byte_path = (fb[0] == '1');
        // Small batches (an equal share would be under ~1000 columns) leave the persistent grid: the mixed-loci mode below,
        // or one workgroup per locus-aligned slice (cutting a small locus in two doubles its prologue and drain: measured on C2).
        p->site_persistent = (ncols / p->site_waves >= 1000) ? 1 : 0;
        if (const char* e3 = getenv("TPHIP_SITE_PERSISTENT")) p->site_persistent = (e3[0] == '1');
        if (!p->site_persistent && p->site_lds_depth < p->prog.stack_depth) {   // the scratch rows are per persistent wave
            p->site_lds_depth = p->prog.stack_depth;
            p->site_waves = per_cu * prop.multiProcessorCount;
        }
        // Small batches of trees up to 64 taxa: equal shares of the work list, a wave carrying columns of several loci at
        // once (site_rate_kernel<NW, false, true>); the slice mode remains for the streamed-words and byte paths.
        p->site_mixed = (!p->site_persistent && p->nwords <= 8 && !p->force_byte_path && ncols < ((int64_t)1 << kMixedColBits)) ? 1 : 0;
        if (const char* em = getenv("TPHIP_SITE_MIXED")) p->site_mixed = (em[0] == '1') && p->nwords <= 8 && !p->force_byte_path && ncols < ((int64_t)1 << kMixedColBits);
        if (p->site_mixed) {
            int per3 = 0;
            const size_t lds3 = (kMixedLdsHeader + (size_t)p->prog.stack_depth * 12 * kSiteBlock) * sizeof(double);
            const int64_t simds = 4 * (int64_t)prop.multiProcessorCount;
            const double est_rounds = (double)ncols * (p->start_rule == TPHIP_START_REFERENCE ? 2.2 : 1.4) / (double)(kSiteBlock * simds);
            if (lds3 > 160 * 1024 || site_rate_kernel_occupancy(kMixedVariant + (p->nwords <= 2 ? 2 : 8), lds3, &per3) != hipSuccess || per3 < 1) p->site_mixed = 0;
            // a batch that wants two waves per SIMD on a tree whose tables leave room for six per CU (64 taxa: 23.3 KB of LDS
            // per wave) stays with the slices: 1900 loci x 1000 x 64, 3.36 ms in slices, 4.10 with 1024 mixed waves
            else if (est_rounds >= 28.0 && per3 < 8 && !getenv("TPHIP_SITE_MIXED")) p->site_mixed = 0;
            else {
                p->site_persistent = 0;
                // One wave per SIMD or two?  A second wave on a SIMD adds half again to its throughput (C2: 15.2 us per round of
                // evaluations alone, 19.8 us each for two) but lengthens every round of the wave that carries the slowest column
                // (22 evaluations on C2 when the average column takes 3.7), and the launch ends with that wave.  Measured, one
                // per SIMD / two: C2 (17 rounds of evaluations per SIMD) 0.348 / 0.391 ms, two C2s (34 rounds) 0.789 / 0.508,
                // 300 loci x 1000 x 64 (8 rounds) 0.757 / 0.932, the resampled 5-taxon locus of bench.py --workload R1 (18 rounds;
                // 1.5 after de-duplication) 0.150 / 0.165 and 0.099 / 0.138; 1152 waves on C2 0.422 (SIMDs shared unevenly).
                // The grid is two per SIMD where eight waves fit a CU (64-taxon trees hold six: one per SIMD, and the batches
                // that want two stay with the slices, above); the kernel uses half of it when the work list -- known on the
                // device only, after classification and de-duplication -- is shorter than ~24 rounds per SIMD at 3.7
                // evaluations per column from HyPhy's start value, 2.4 from the parsimony start.
                const int64_t want = (per3 < 8) ? std::min<int64_t>(simds, (int64_t)per3 * prop.multiProcessorCount)
                                                : 8 * (int64_t)prop.multiProcessorCount;
                p->mixed_few_waves = (int32_t)std::min<int64_t>(simds, want);
                p->mixed_switch_cols = (int64_t)(24.0 * (double)(kSiteBlock * simds) / (p->start_rule == TPHIP_START_REFERENCE ? 3.7 : 2.4));
                p->site_waves = (int32_t)std::max<int64_t>(1, std::min<int64_t>(want, (ncols + kSiteBlock - 1) / kSiteBlock));
            }
        }
        if (const char* e2 = getenv("TPHIP_SITE_WAVES")) { long v = atol(e2); if (v >= 1) { p->site_waves = (int32_t)v; p->mixed_switch_cols = 0; } }  // tuning knob
        // Share sizes of the persistent grid.  Equal shares (one per resident wave) are equal column counts, not equal
        // work: loci differ in evaluations per column, and with 5-7 resident waves per CU (deep LDS stacks) a wave that
        // shares its SIMD runs slower than one that does not.  So the resident waves' first shares take 80-90 % of the
        // batch and the rest is cut into twice as many small shares that the dispatcher hands to whichever waves
        // finish first: site_rate_kernel C3 8.4 -> 7.0 ms, C4 share 9.8 -> 9.2, C5 share 20.9 -> 17.1.  Batches whose
        // shares are short anyway only gain drains (2.2 M columns in 2200 loci: +3.5 %): they keep equal shares.
        {
            const int64_t share = ncols / std::max(1, p->site_waves);
            const bool uneven = p->site_persistent && share >= 1500;
            p->site_grid_mult = uneven ? 3 : 1;
            // a share that spans several short loci pays a drain at every locus boundary, so its tail shares should be
            // fewer columns: 90 % up front there (C4: 9.0 vs 9.2-9.3 ms), 80 % when shares lie inside long loci (C3: 6.9 vs 7.0)
            const int64_t avg_locus = ncols / std::max<int64_t>(1, d->nloci);
            p->site_first_fraction = !uneven ? 0.0 : (share >= 2 * avg_locus ? 0.9 : 0.8);
            // whole C5 with the spilled stack slot (8 waves per CU): (3, 0.8) 109.6 ms, (3, 0.9) 113.3, (4, 0.85) 109.1, (2, 0.8) 117.1
            if (uneven && p->site_lds_depth < p->prog.stack_depth) p->site_first_fraction = 0.8;
            if (const char* e8 = getenv("TPHIP_SITE_FIRST_FRACTION")) p->site_first_fraction = atof(e8);
            if (const char* e7 = getenv("TPHIP_SITE_GRID_MULT")) { long v = atol(e7); if (v >= 1 && v <= 16) p->site_grid_mult = (int32_t)v; }
        }
    }
    // workspace layout
    size_t off = 0;
    p->ws_work_cols = off; off = align_up(off + sizeof(int32_t) * (size_t)ncols, 256);
    if (!p->site_persistent && !getenv("TPHIP_SITE_NO_REORDER")) {   // small batches: slow-columns-first copy of the work list
        p->ws_work_cols2 = off; off = align_up(off + sizeof(int32_t) * (size_t)ncols, 256);
    }
    p->ws_work_count = off; off = align_up(off + sizeof(int32_t) * (size_t)d->nloci, 256);
    p->ws_work_prefix = off; off = align_up(off + sizeof(int64_t) * ((size_t)d->nloci + 1), 256);
    p->ws_slice_prefix = off; off = align_up(off + sizeof(int64_t) * ((size_t)d->nloci + 1), 256);
    p->ws_partial = off; off = align_up(off + sizeof(double) * (size_t)p->n_pi_chunks * (size_t)(d->T + 2 * d->n_i), 256);
    p->ws_packed = off; off = align_up(off + sizeof(uint32_t) * (size_t)p->nwords * (size_t)ncols, 256);
    if (const char* e9 = getenv("TPHIP_DEDUP")) {   // test/tuning knob: 0 = never, 1 = always, anything else = automatic
        p->dedup_mode = (e9[0] == '0') ? DEDUP_OFF : (e9[0] == '1') ? DEDUP_ON : DEDUP_AUTO;
    }
    if (p->dedup_mode == DEDUP_AUTO && ncols < kDedupAutoMinColumns) p->dedup_mode = DEDUP_OFF;   // small batch: see pattern_kernels.hpp
    if (p->dedup_mode != DEDUP_OFF) {
        p->ws_hash = off; off = align_up(off + sizeof(uint64_t) * (size_t)ncols, 256);
        p->ws_dup_of = off; off = align_up(off + sizeof(int32_t) * (size_t)ncols, 256);
        p->ws_tab_key = off; off = align_up(off + sizeof(unsigned long long) * 2 * (size_t)ncols, 256);
        p->ws_tab_val = off; off = align_up(off + sizeof(int32_t) * 2 * (size_t)ncols, 256);
        p->ws_dedup_on = off; off = align_up(off + sizeof(int32_t) * (size_t)d->nloci, 256);
    }
    if (p->site_lds_depth < p->prog.stack_depth) {
        const size_t rows = (size_t)p->site_waves * (size_t)p->site_grid_mult * (size_t)(p->prog.stack_depth - p->site_lds_depth);
        p->ws_spill = off; off = align_up(off + rows * 12 * kSiteBlock * sizeof(double), 256);
    }
    p->ws_total = off + 256;
    p->saved = save_desc(d);
    p->host_split = 2;   // measured (C3 / C4, pinned buffers): 1 group 16.5 / 151 ms, 2: 12.8 / 127, 3: 17.3 / 154, 4: 14.9 / 140, 6: 15.2 / 135
    if (const char* hs = getenv("TPHIP_HOST_SPLIT")) p->host_split = std::max(1, std::min(64, atoi(hs)));
    *out = p;
    return TPHIP_OK;
}

int32_t tphip_plan_table_width(const tphip_plan* p) { return p ? p->T + p->n_t + 2 * p->n_i : 0; }
int64_t tphip_plan_ncols(const tphip_plan* p) { return p ? p->ncols : 0; }
size_t tphip_plan_workspace_bytes(const tphip_plan* p) { return p ? p->ws_total : 0; }
double tphip_plan_chrono_length(const tphip_plan* p) { return p ? p->prog.chrono_length : 0.0; }
int32_t tphip_plan_stack_depth(const tphip_plan* p) { return p ? p->prog.stack_depth : 0; }

int tphip_plan_op_counts(const tphip_plan* p, int32_t* counts) {
    if (!p || !counts) return fail(TPHIP_ERR_INVALID, "null argument");
    for (int i = 0; i < 5; ++i) counts[i] = 0;
    for (const TreeOp& op : p->prog.ops) ++counts[op.code];
    return TPHIP_OK;
}

int32_t tphip_plan_cherry_count(const tphip_plan* p) {
    if (!p) return 0;
    int32_t n = 0;
    for (const TreeOp& op : p->prog.fused_ops) n += ((op.code & OP_CODE_MASK) == OP_CHERRY);
    return n;
}

int tphip_plan_get_models(const tphip_plan* p, double* lam, double* U, double* Uinv, double* kappa) {
    if (!p) return fail(TPHIP_ERR_INVALID, "null plan");
    HIP_TRY(hipSetDevice(p->device));
    std::vector<LocusModel> h((size_t)p->nloci);
    HIP_TRY(hipMemcpy(h.data(), p->d_models.p, sizeof(LocusModel) * h.size(), hipMemcpyDeviceToHost));
    for (int64_t l = 0; l < p->nloci; ++l) {  // expand the packed device layout to full 4x4 matrices
        const LocusModel& m = h[l];
        if (lam) { lam[4 * l] = 0.0; for (int k = 1; k < 4; ++k) lam[4 * l + k] = m.lam[k - 1]; }
        if (U) for (int i = 0; i < 4; ++i) { U[16 * l + 4 * i] = 1.0; for (int k = 1; k < 4; ++k) U[16 * l + 4 * i + k] = m.U[i * 3 + k - 1]; }
        if (Uinv) for (int j = 0; j < 4; ++j) { Uinv[16 * l + j] = m.pi[j]; for (int k = 1; k < 4; ++k) Uinv[16 * l + 4 * k + j] = m.Ui[(k - 1) * 4 + j]; }
        if (kappa) kappa[l] = m.kappa;
    }
    return TPHIP_OK;
}

int tphip_profile_enable(tphip_plan* p, int32_t on) {
    if (!p) return fail(TPHIP_ERR_INVALID, "null plan");
    HIP_TRY(hipSetDevice(p->device));
    if (on && p->ev.empty()) {
        p->ev.resize(4 * kProfileRing);
        for (auto& e : p->ev) HIP_TRY(hipEventCreate(&e));
    }
    p->profile = on != 0;
    return TPHIP_OK;
}

static int profile_drain(tphip_plan* p) {
    for (int s = 0; s < p->ev_used; ++s) {
        float ms = 0;
        HIP_TRY(hipEventSynchronize(p->ev[4 * s + 1]));
        HIP_TRY(hipEventElapsedTime(&ms, p->ev[4 * s + 0], p->ev[4 * s + 1]));
        p->acc_site_ms += ms;
        HIP_TRY(hipEventSynchronize(p->ev[4 * s + 3]));
        HIP_TRY(hipEventElapsedTime(&ms, p->ev[4 * s + 2], p->ev[4 * s + 3]));
        p->acc_pi_ms += ms;
        ++p->acc_launches;
    }
    p->ev_used = 0;
    return TPHIP_OK;
}

int tphip_profile_read(tphip_plan* p, double* site_ms, double* pi_ms, int64_t* launches, int32_t reset) {
    if (!p) return fail(TPHIP_ERR_INVALID, "null plan");
    HIP_TRY(hipSetDevice(p->device));
    int rc = profile_drain(p);
    if (rc) return rc;
    if (site_ms) *site_ms = p->acc_site_ms;
    if (pi_ms) *pi_ms = p->acc_pi_ms;
    if (launches) *launches = p->acc_launches;
    if (reset) { p->acc_site_ms = p->acc_pi_ms = 0; p->acc_launches = 0; }
    return TPHIP_OK;
}

int tphip_last_eval_count(tphip_plan* p, int64_t* evals) {
    if (!p || !evals) return fail(TPHIP_ERR_INVALID, "null argument");
    HIP_TRY(hipSetDevice(p->device));
    unsigned long long v = 0;
    if (p->last_run_in_parts) {   // the last site-rate launch was a host-pointer pipeline of locus groups: their sum
        for (tphip_plan* q : p->parts) {
            unsigned long long w = 0;
            HIP_TRY(hipMemcpy(&w, q->d_evals.p, sizeof w, hipMemcpyDeviceToHost));
            v += w;
        }
    } else {
        HIP_TRY(hipMemcpy(&v, p->d_evals.p, sizeof v, hipMemcpyDeviceToHost));
        if (getenv("TPHIP_SITE_TRACE_ROUNDS")) {   // meaningful in a -DTPHIP_SITE_TRACE_ROUNDS build only
            unsigned long long c[8];
            HIP_TRY(hipMemcpy(c, p->d_evals.p, sizeof c, hipMemcpyDeviceToHost));
            if (c[5]) fprintf(stderr, "site_rate_kernel: %llu waves with work, evaluations %llu, rounds mean %.1f max %llu (lane use %.1f %%), wave time mean %.1f us max %.1f us\n",
                              c[5], c[0], (double)c[1] / c[5], c[2], 100.0 * c[0] / (64.0 * c[1]), 0.01 * c[3] / c[5], 0.01 * c[4]);
        }
    }
    *evals = (int64_t)v;
    return TPHIP_OK;
}

// ---- launches ------------------------------------------------------------------------------------

static PiParams pi_params(const tphip_plan* p, const double* d_rates, const int32_t* d_nres, void* ws);

static int launch_site_rates(tphip_plan* p, const uint8_t* d_states, double* d_rate, double* d_subst, double* d_lnl,
                             uint8_t* d_flag, int32_t* d_nres, void* ws, hipStream_t st, int slot) {
    p->last_run_in_parts = false;
    int32_t* work_cols = (int32_t*)((char*)ws + p->ws_work_cols);
    int32_t* work_count = (int32_t*)((char*)ws + p->ws_work_count);
    ClassifyParams C;
    C.states = d_states; C.ncols_total = p->ncols; C.ntaxa = p->ntaxa; C.models = p->d_models.p;
    C.locus_offsets = p->d_offsets.p; C.chunk_locus = p->d_pi_chunk_locus.p; C.chunk_index = p->d_pi_chunk_index.p;
    C.rate = d_rate; C.subst = d_subst; C.lnl = d_lnl; C.flag = d_flag; C.nres = d_nres;
    C.chrono_length = p->prog.chrono_length;
    C.ops = p->d_ops.p; C.nops = (int32_t)p->prog.ops.size();
    C.packed = (uint32_t*)((char*)ws + p->ws_packed);
    C.tip_taxon = p->d_tip_taxon.p;
    C.start_scale = (p->start_rule == TPHIP_START_REFERENCE) ? 0.0 : 1.0;
    const bool dedup = p->dedup_mode != DEDUP_OFF && p->n_pi_chunks > 0;
    C.hash = dedup ? (uint64_t*)((char*)ws + p->ws_hash) : nullptr;
    DedupParams D;
    if (dedup) {
        D.locus_offsets = p->d_offsets.p; D.chunk_locus = p->d_pi_chunk_locus.p; D.chunk_index = p->d_pi_chunk_index.p;
        D.hash = C.hash; D.packed = C.packed; D.nwords = p->nwords; D.ncols_total = p->ncols; D.flag = d_flag;
        D.dup_of = (int32_t*)((char*)ws + p->ws_dup_of);
        D.tab_key = (unsigned long long*)((char*)ws + p->ws_tab_key); D.tab_val = (int32_t*)((char*)ws + p->ws_tab_val);
        D.on = (int32_t*)((char*)ws + p->ws_dedup_on); D.mode = p->dedup_mode;
        D.rate = d_rate; D.subst = d_subst; D.lnl = d_lnl;
    }
    if (p->n_pi_chunks > 0) {
        classify_kernel<<<dim3((unsigned)p->n_pi_chunks), dim3(kPiBlock), 0, st>>>(C);
        if (dedup) {   // one rate per unique site pattern (bf:1033-1044); loci that hardly repeat a column skip it
            if (p->max_locus_cols > 2048) dedup_estimate_kernel<1024><<<dim3((unsigned)p->nloci), dim3(1024), 0, st>>>(D);
            else dedup_estimate_kernel<256><<<dim3((unsigned)p->nloci), dim3(256), 0, st>>>(D);
            dedup_insert_kernel<<<dim3((unsigned)p->n_pi_chunks), dim3(256), 0, st>>>(D);
            dedup_resolve_kernel<<<dim3((unsigned)p->n_pi_chunks), dim3(256), 0, st>>>(D);
        }
        if (p->max_locus_cols > 2048) compact_kernel<1024><<<dim3((unsigned)p->nloci), dim3(1024), 0, st>>>(d_flag, p->d_offsets.p, work_cols, work_count);
        else compact_kernel<256><<<dim3((unsigned)p->nloci), dim3(256), 0, st>>>(d_flag, p->d_offsets.p, work_cols, work_count);
        HIP_TRY(launch_scan_counts_kernel(st, work_count, p->nloci, p->site_chunk_cols, (int64_t*)((char*)ws + p->ws_work_prefix),
                                          (int64_t*)((char*)ws + p->ws_slice_prefix)));
    }
    HIP_TRY(hipMemsetAsync(p->d_evals.p, 0, 8 * sizeof(unsigned long long), st));
    SiteParams S;
    S.states = d_states; S.ncols_total = p->ncols; S.models = p->d_models.p; S.ops = p->d_ops.p;
    S.nops = (int32_t)p->prog.ops.size(); S.stack_depth = p->prog.stack_depth; S.chrono_length = p->prog.chrono_length;
    S.locus_offsets = p->d_offsets.p; S.chunk_locus = p->d_site_chunk_locus.p; S.chunk_index = p->d_site_chunk_index.p;
    S.chunk_cols = p->site_chunk_cols;
    S.packed = (const uint32_t*)((char*)ws + p->ws_packed); S.nwords = p->nwords;
    S.work_cols = work_cols; S.work_count = work_count;
    S.work_cols2 = (!p->site_persistent && p->ws_work_cols2) ? (int32_t*)((char*)ws + p->ws_work_cols2) : nullptr;
    S.work_prefix = (const int64_t*)((char*)ws + p->ws_work_prefix); S.nloci = p->nloci;
    S.slice_prefix = (const int64_t*)((char*)ws + p->ws_slice_prefix);
    S.rate = d_rate; S.subst = d_subst; S.lnl = d_lnl; S.flag = d_flag; S.eval_counter = p->d_evals.p;
    const bool spill = p->site_lds_depth < p->prog.stack_depth;
    const size_t lds = (kSiteLdsHeader + (size_t)p->site_lds_depth * 12 * kSiteBlock) * sizeof(double);
    S.lds_depth = p->site_lds_depth;
    S.spill = spill ? (double*)((char*)ws + p->ws_spill) : nullptr;
    S.persistent = (p->site_persistent || p->site_mixed) ? 1 : 0;
    S.first_round = p->site_waves;
    S.mixed_few_waves = p->mixed_few_waves; S.mixed_switch_cols = p->mixed_switch_cols;
    S.first_fraction = (p->site_first_fraction > 0.0) ? p->site_first_fraction : 1.0 / (double)p->site_grid_mult;
    S.ncat = p->ncat; S.cat = p->d_cat.p;
    // profiling brackets exactly the dominant kernel, so the figure matches rocprofv3's per-kernel average
    if (slot >= 0) HIP_TRY(hipEventRecord(p->ev[4 * slot + 0], st));
    if (p->n_site_chunks > 0) {
        const bool mixed = p->site_mixed && !p->force_byte_path;
        const dim3 grid((unsigned)(mixed ? p->site_waves : p->site_persistent ? p->site_waves * p->site_grid_mult : p->n_site_chunks));
        // packed tip states: in registers up to 64 tips, streamed one word ahead beyond (site_rate_kernel.hpp)
        const bool byte_path = p->force_byte_path;   // test/tuning knob, resolved at plan creation
        if (!byte_path) {  // the packed path reads the stream with fused cherries (tree_program.hpp)
            S.ops = p->d_fused_ops.p;
            S.nops = (int32_t)p->prog.fused_ops.size();
        }
        // variants: byte path; packed words in registers (<= 16 / <= 64 tips); streamed words (more than 64 tips)
        const size_t lds_full = (kSiteLdsHeader + (size_t)p->prog.stack_depth * 12 * kSiteBlock) * sizeof(double);
        const int variant = byte_path ? 0 : p->nwords <= 2 ? 2 : p->nwords <= 8 ? 8 : (spill ? kStreamWordsSpill : kStreamWords);
        if (byte_path) { S.lds_depth = p->prog.stack_depth; S.spill = nullptr; }
        if (mixed) {   // equal shares: first_round = grid
            const size_t lds_mixed = (kMixedLdsHeader + (size_t)p->prog.stack_depth * 12 * kSiteBlock) * sizeof(double);
            S.first_fraction = 1.0;
            HIP_TRY(launch_site_rate_kernel(kMixedVariant + variant, grid, lds_mixed, st, S));
        } else
        HIP_TRY(launch_site_rate_kernel(variant, grid, (byte_path || !spill) ? lds_full : lds, st, S));
    }
    if (slot >= 0) HIP_TRY(hipEventRecord(p->ev[4 * slot + 1], st));
    if (dedup) dedup_scatter_kernel<<<dim3((unsigned)p->n_pi_chunks), dim3(256), 0, st>>>(D);
    HIP_TRY(hipGetLastError());
    return TPHIP_OK;
}

static int launch_pi_tables(tphip_plan* p, const double* d_rates, const int32_t* d_nres, double* d_tables, void* ws,
                            hipStream_t st, int slot) {
    PiParams Q = pi_params(p, d_rates, d_nres, ws);
    if (slot >= 0) HIP_TRY(hipEventRecord(p->ev[4 * slot + 2], st));
    if (p->n_pi_chunks > 0) pi_partial_kernel<<<dim3((unsigned)p->n_pi_chunks), dim3(kPiBlock), 0, st>>>(Q);
    pi_reduce_kernel<<<dim3((unsigned)p->nloci), dim3(64), 0, st>>>(Q.partial, p->d_locus_pichunk_offsets.p, p->T,
                                                                   p->d_times.p, p->n_t, p->n_i, d_tables);
    if (slot >= 0) HIP_TRY(hipEventRecord(p->ev[4 * slot + 3], st));
    HIP_TRY(hipGetLastError());
    return TPHIP_OK;
}

static int take_slot(tphip_plan* p) {
    if (!p->profile) return -1;
    if (p->ev_used >= kProfileRing) {
        if (profile_drain(p)) return -1;
    }
    return p->ev_used++;
}

int tphip_run_dev(tphip_plan* p, const uint8_t* d_states, double* d_rate, double* d_subst, double* d_lnl, uint8_t* d_flag,
                  int32_t* d_nres, double* d_tables, void* ws, size_t ws_bytes, void* stream) {
    if (!p) return fail(TPHIP_ERR_INVALID, "null plan");
    if (!d_states || !d_rate || !d_subst || !d_lnl || !d_flag || !d_nres || !d_tables || !ws)
        return fail(TPHIP_ERR_INVALID, "null device pointer");
    if (ws_bytes < p->ws_total) return fail(TPHIP_ERR_WORKSPACE, "workspace smaller than tphip_plan_workspace_bytes()");
    HIP_TRY(hipSetDevice(p->device));
    hipStream_t st = (hipStream_t)stream;
    int slot = take_slot(p);
    int rc = launch_site_rates(p, d_states, d_rate, d_subst, d_lnl, d_flag, d_nres, ws, st, slot);
    if (rc) return rc;
    return launch_pi_tables(p, d_rate, d_nres, d_tables, ws, st, slot);
}

int tphip_site_rates_dev(tphip_plan* p, const uint8_t* d_states, double* d_rate, double* d_subst, double* d_lnl,
                         uint8_t* d_flag, int32_t* d_nres, void* ws, size_t ws_bytes, void* stream) {
    if (!p) return fail(TPHIP_ERR_INVALID, "null plan");
    if (!d_states || !d_rate || !d_subst || !d_lnl || !d_flag || !d_nres || !ws)
        return fail(TPHIP_ERR_INVALID, "null device pointer");
    if (ws_bytes < p->ws_total) return fail(TPHIP_ERR_WORKSPACE, "workspace smaller than tphip_plan_workspace_bytes()");
    HIP_TRY(hipSetDevice(p->device));
    int slot = take_slot(p);
    int rc = launch_site_rates(p, d_states, d_rate, d_subst, d_lnl, d_flag, d_nres, ws, (hipStream_t)stream, slot);
    if (rc == 0 && slot >= 0) {  // keep the pi event pair of this slot valid (zero-length)
        HIP_TRY(hipEventRecord(p->ev[4 * slot + 2], (hipStream_t)stream));
        HIP_TRY(hipEventRecord(p->ev[4 * slot + 3], (hipStream_t)stream));
    }
    return rc;
}

int tphip_pi_tables_dev(tphip_plan* p, const double* d_rates, const int32_t* d_nres, double* d_tables, void* ws,
                        size_t ws_bytes, void* stream) {
    if (!p) return fail(TPHIP_ERR_INVALID, "null plan");
    if (!d_rates || !d_tables || !ws) return fail(TPHIP_ERR_INVALID, "null device pointer");
    if (ws_bytes < p->ws_total) return fail(TPHIP_ERR_WORKSPACE, "workspace smaller than tphip_plan_workspace_bytes()");
    HIP_TRY(hipSetDevice(p->device));
    int slot = take_slot(p);
    if (slot >= 0) {
        HIP_TRY(hipEventRecord(p->ev[4 * slot + 0], (hipStream_t)stream));
        HIP_TRY(hipEventRecord(p->ev[4 * slot + 1], (hipStream_t)stream));
    }
    return launch_pi_tables(p, d_rates, d_nres, d_tables, ws, (hipStream_t)stream, slot);
}

static PiParams pi_params(const tphip_plan* p, const double* d_rates, const int32_t* d_nres, void* ws) {
    PiParams Q;
    Q.rates = d_rates; Q.nres = d_nres; Q.locus_offsets = p->d_offsets.p;
    Q.chunk_locus = p->d_pi_chunk_locus.p; Q.chunk_index = p->d_pi_chunk_index.p;
    Q.T = p->T; Q.intervals = p->d_intervals.p; Q.n_i = p->n_i; Q.integ_mode = p->integ_mode;
    Q.correction = p->correction; Q.threshold = p->threshold;
    Q.round_scale = (p->round_decimals >= 0) ? std::pow(10.0, (double)p->round_decimals) : 0.0;
    Q.partial = ws ? (double*)((char*)ws + p->ws_partial) : nullptr;
    return Q;
}

int tphip_corrected_rates_dev(tphip_plan* p, const double* d_rates, const int32_t* d_nres, double* d_out, void* stream) {
    if (!p) return fail(TPHIP_ERR_INVALID, "null plan");
    if (p->ncols && (!d_rates || !d_out)) return fail(TPHIP_ERR_INVALID, "null device pointer");
    HIP_TRY(hipSetDevice(p->device));
    if (p->ncols) {
        corrected_rates_kernel<<<dim3((unsigned)((p->ncols + 255) / 256)), dim3(256), 0, (hipStream_t)stream>>>(
            pi_params(p, d_rates, d_nres, nullptr), p->ncols, d_out);
        HIP_TRY(hipGetLastError());
    }
    return TPHIP_OK;
}

int tphip_corrected_rates(tphip_plan* p, const double* rates, const int32_t* nres, double* out) {
    if (!p) return fail(TPHIP_ERR_INVALID, "null plan");
    if (p->ncols == 0) return TPHIP_OK;
    if (!rates || !out) return fail(TPHIP_ERR_INVALID, "null host pointer");
    HIP_TRY(hipSetDevice(p->device));
    Scratch S;
    const size_t n = (size_t)p->ncols;
    double* d_r = S.get<double>(n);
    double* d_o = S.get<double>(n);
    int32_t* d_n = nres ? S.get<int32_t>(n) : nullptr;
    if (!d_r || !d_o || (nres && !d_n)) return fail(TPHIP_ERR_HIP, "hipMalloc failed");
    HIP_TRY(hipMemcpy(d_r, rates, sizeof(double) * n, hipMemcpyHostToDevice));
    if (nres) HIP_TRY(hipMemcpy(d_n, nres, sizeof(int32_t) * n, hipMemcpyHostToDevice));
    int rc = tphip_corrected_rates_dev(p, d_r, d_n, d_o, nullptr);
    if (rc) return rc;
    HIP_TRY(hipMemcpy(out, d_o, sizeof(double) * n, hipMemcpyDeviceToHost));
    return TPHIP_OK;
}

int tphip_townsend_pi_dense_dev(int32_t device, const double* d_rates, int64_t n, const double* d_times, int32_t n_times,
                                double* d_out, void* stream) {
    if (tphip_device_count() <= 0) return fail(TPHIP_ERR_NO_DEVICE, "no HIP device visible: libtphip has no CPU path");
    if (n < 0 || n_times < 0 || (n && n_times && (!d_rates || !d_out || !d_times))) return fail(TPHIP_ERR_INVALID, "bad arguments");
    HIP_TRY(hipSetDevice(device));
    if (n && n_times) {
        townsend_dense_kernel<<<dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream>>>(d_rates, n, d_times,
                                                                                                    n_times, d_out);
        HIP_TRY(hipGetLastError());
    }
    return TPHIP_OK;
}

int tphip_quad_townsend_dev(int32_t device, const double* d_rates, int64_t n, double a, double b, int32_t integ_mode,
                            double* d_integral, double* d_abserr, void* stream) {
    if (tphip_device_count() <= 0) return fail(TPHIP_ERR_NO_DEVICE, "no HIP device visible: libtphip has no CPU path");
    if (n < 0 || (n && (!d_rates || !d_integral || !d_abserr))) return fail(TPHIP_ERR_INVALID, "bad arguments");
    if (!(a < b)) return fail(TPHIP_ERR_INVALID, "Start time is sooner than end time in an interval");
    if (integ_mode != TPHIP_INTEG_QUADPACK && integ_mode != TPHIP_INTEG_CLOSED) return fail(TPHIP_ERR_INVALID, "unknown integ_mode");
    HIP_TRY(hipSetDevice(device));
    if (n) {
        quad_sites_kernel<<<dim3((unsigned)((n + 127) / 128)), dim3(128), 0, (hipStream_t)stream>>>(d_rates, n, a, b, integ_mode,
                                                                                                d_integral, d_abserr);
        HIP_TRY(hipGetLastError());
    }
    return TPHIP_OK;
}

// Slices per candidate for the locus likelihood / gradient kernels: enough work items to fill the device even when
// only a few candidates are in flight (the general model's one point per locus), never more than a locus has blocks.
static int lik_nsplit(const tphip_plan* p, int64_t ncand, int block) {
    if (p->lik_nsplit_forced > 0) return p->lik_nsplit_forced;
    const int64_t target = (int64_t)p->num_cus * 8;
    int64_t ns = (target + ncand - 1) / std::max<int64_t>(ncand, 1);
    const int64_t max_blocks = std::max<int64_t>(1, (p->max_locus_cols + block - 1) / block);
    ns = std::min(ns, max_blocks);
    return (int)std::max<int64_t>(1, std::min<int64_t>(ns, 4096));
}

static int grow_part(tphip_plan* p, size_t need) {
    if (need <= p->part_bytes) return TPHIP_OK;
    if (p->d_part) { HIP_TRY(hipFree(p->d_part)); p->d_part = nullptr; p->part_bytes = 0; }
    HIP_TRY(hipMalloc((void**)&p->d_part, need));
    p->part_bytes = need;
    return TPHIP_OK;
}

int tphip_locus_loglik_dev(tphip_plan* p, const uint8_t* d_states, int64_t ncand, const int32_t* d_cand_locus,
                           const double* d_cand_exch, const double* d_blen_vecs, const int32_t* d_cand_vec,
                           const double* d_cand_scale, const int32_t* d_cand_pidx, const double* d_cand_pfac, double* d_out,
                           void* stream) {
    if (!p) return fail(TPHIP_ERR_INVALID, "null plan");
    if (ncand < 0 || (ncand && (!d_states || !d_cand_locus || !d_cand_exch || !d_blen_vecs || !d_cand_vec || !d_cand_scale ||
                                !d_cand_pidx || !d_cand_pfac || !d_out)))
        return fail(TPHIP_ERR_INVALID, "null device pointer");
    if (ncand == 0) return TPHIP_OK;
    HIP_TRY(hipSetDevice(p->device));
    LikParams L;
    L.states = d_states; L.ncols_total = p->ncols; L.locus_offsets = p->d_offsets.p; L.models = p->d_models.p;
    L.col_weight = p->d_col_weight;
    L.lops = p->d_lik_ops.p; L.nops = (int32_t)p->prog.ops.size(); L.nnodes = p->nnodes; L.ntaxa = p->ntaxa;
    L.stack_depth = p->prog.stack_depth; L.cand_locus = d_cand_locus; L.cand_exch = d_cand_exch;
    L.blen_vecs = d_blen_vecs; L.cand_vec = d_cand_vec; L.cand_scale = d_cand_scale; L.cand_pidx = d_cand_pidx;
    L.cand_pfac = d_cand_pfac; L.out = d_out;
    if (p->value_cols > 0) {
        // transition-matrix form: eigen-systems and matrices of a chunk of candidates, then the pruning kernel
        const int C = p->value_cols;
        const int nsplit = lik_nsplit(p, ncand, kLikBlock * C);
        if (nsplit > 1) {
            int rc = grow_part(p, (size_t)ncand * nsplit * sizeof(double));
            if (rc) return rc;
        }
        const size_t per_cand = ((size_t)p->nnodes * 16 + 36) * sizeof(double);
        int64_t chunk = std::max<int64_t>(1, (int64_t)(kValueWorkspaceBytes / per_cand));
        chunk = std::min<int64_t>(chunk, std::max<int64_t>(1, ((int64_t)1 << 30) / nsplit));   // grid.x limit
        chunk = std::min<int64_t>(chunk, ncand);
        const size_t need = (size_t)chunk * per_cand;
        if (need > p->value_ws_bytes) {
            if (p->d_value_ws) { HIP_TRY(hipFree(p->d_value_ws)); p->d_value_ws = nullptr; p->value_ws_bytes = 0; }
            HIP_TRY(hipMalloc((void**)&p->d_value_ws, need));
            p->value_ws_bytes = need;
        }
        double* d_eig = p->d_value_ws;
        double* d_pmat = p->d_value_ws + (size_t)chunk * 36;
        ValueParams V;
        V.states = d_states; V.ncols_total = p->ncols; V.locus_offsets = p->d_offsets.p; V.col_weight = p->d_col_weight;
        V.models = p->d_models.p; V.vops = p->d_value_ops.p; V.nvops = p->value_nops; V.ntaxa = p->ntaxa;
        V.nnodes = p->nnodes; V.pmat = d_pmat; V.nsplit = nsplit;
        V.tip_taxon = p->d_tip_taxon.p; V.nwords = p->nwords; V.tip_node = p->d_value_tip_node.p;
        V.packed = (d_states == p->lib_states) ? p->d_value_packed : nullptr;
        hipStream_t st = (hipStream_t)stream;
        for (int64_t done = 0; done < ncand; done += chunk) {
            const int64_t n = std::min<int64_t>(ncand - done, chunk);
            HIP_TRY(launch_lik_eigen_kernel(st, p->d_models.p, d_cand_locus + done, d_cand_exch + done * 6, n, d_eig));
            HIP_TRY(launch_lik_pmat_kernel(st, d_eig, d_blen_vecs, d_cand_vec + done, d_cand_scale + done, d_cand_pidx + done,
                                           d_cand_pfac + done, n, p->nnodes, d_pmat));
            V.cand_locus = d_cand_locus + done;
            V.out = (nsplit > 1 ? p->d_part : d_out) + done * nsplit;
            HIP_TRY(launch_locus_value_kernel(C, p->prog.stack_depth, dim3((unsigned)(n * nsplit)), p->value_lds, st, V));
        }
        HIP_TRY(hipGetLastError());
        if (nsplit > 1) {
            HIP_TRY(launch_split_sum_kernel(st, p->d_part, d_out, ncand, nsplit, 1));
        }
        return TPHIP_OK;
    }
    if (!p->lik_ok) return fail(TPHIP_ERR_INVALID, "tree too large for the locus-likelihood kernel's LDS tables");
    const size_t lds = p->lik_lds;
    L.stage_states = p->lik_stage;
    const int nsplit = lik_nsplit(p, ncand, kLikBlock);
    L.nsplit = nsplit;
    if (nsplit > 1) {
        int rc = grow_part(p, (size_t)ncand * nsplit * sizeof(double));
        if (rc) return rc;
        L.out = p->d_part;
    }
    const int64_t per_launch = std::max<int64_t>(1, ((int64_t)1 << 30) / nsplit);   // grid.x limit
    for (int64_t done = 0; done < ncand; done += per_launch) {
        const int64_t n = std::min<int64_t>(ncand - done, per_launch);
        LikParams Q = L;
        Q.cand_locus += done; Q.cand_exch += done * 6; Q.cand_vec += done; Q.cand_scale += done; Q.cand_pidx += done;
        Q.cand_pfac += done; Q.out += done * nsplit;
        HIP_TRY(launch_locus_loglik_kernel(dim3((unsigned)(n * nsplit)), lds, (hipStream_t)stream, Q));
    }
    HIP_TRY(hipGetLastError());
    if (nsplit > 1) {
        HIP_TRY(launch_split_sum_kernel((hipStream_t)stream, p->d_part, d_out, ncand, nsplit, 1));
    }
    return TPHIP_OK;
}

int tphip_locus_gradient_dev(tphip_plan* p, const uint8_t* d_states, int64_t ncand, const int32_t* d_cand_locus,
                             const double* d_cand_exch, const double* d_blen_vecs, const int32_t* d_cand_vec,
                             const double* d_cand_scale, const int32_t* d_cand_pidx, const double* d_cand_pfac, double* d_lnl,
                             double* d_dexch, double* d_dlogt, double* d_sum_dlogt, double* d_d2logt, void* stream) {
    if (!p) return fail(TPHIP_ERR_INVALID, "null plan");
    if (ncand < 0 || (ncand && (!d_states || !d_cand_locus || !d_cand_exch || !d_blen_vecs || !d_cand_vec || !d_cand_scale ||
                                !d_cand_pidx || !d_cand_pfac || !d_lnl || !d_dexch || !d_sum_dlogt)))
        return fail(TPHIP_ERR_INVALID, "null device pointer");
    if (ncand == 0) return TPHIP_OK;
    HIP_TRY(hipSetDevice(p->device));
    if (p->grad2_ok && d_states == p->lib_states && p->d_value_packed) {
        // transition-matrix gradient kernel (locus_grad2_kernel.hpp): eigen-systems, matrices and branch tables of a chunk
        // of candidates, then resident workgroups loop over (candidate, column slice) items, each with its own tape
        hipStream_t st = (hipStream_t)stream;
        const int nsplit = lik_nsplit(p, ncand, kGrad2Block);
        const size_t nn = (size_t)p->nnodes;
        const size_t per_cand = (36 + nn * (16 + kGrad2EF)) * sizeof(double);
        int64_t chunk = std::max<int64_t>(1, (int64_t)(((size_t)4 << 30) / per_cand));
        chunk = std::min<int64_t>(chunk, ncand);
        const size_t need_ws = (size_t)chunk * per_cand;
        if (need_ws > p->grad2_ws_bytes) {
            if (p->d_grad2_ws) { HIP_TRY(hipFree(p->d_grad2_ws)); p->d_grad2_ws = nullptr; p->grad2_ws_bytes = 0; }
            HIP_TRY(hipMalloc((void**)&p->d_grad2_ws, need_ws));
            p->grad2_ws_bytes = need_ws;
        }
        double* d_eig = p->d_grad2_ws;
        double* d_pmat = d_eig + (size_t)chunk * 36;
        double* d_ef = d_pmat + (size_t)chunk * nn * 16;
        const size_t items_all = (size_t)ncand * nsplit;
        if (nsplit > 1) {
            int rc = grow_part(p, items_all * (8 + (d_dlogt ? nn : 0) + (d_d2logt ? nn : 0)) * sizeof(double));
            if (rc) return rc;
        }
        const int64_t grid_max = (int64_t)p->num_cus * p->grad2_blocks_per_cu;
        const size_t need_tape = (size_t)std::min<int64_t>((int64_t)std::min<int64_t>(chunk, ncand) * nsplit, grid_max) *
                                 (size_t)std::max(1, p->grad2_ntape) * 4 * kGrad2Block * sizeof(double);
        if (need_tape > p->tape_bytes) {
            if (p->d_tape) { HIP_TRY(hipFree(p->d_tape)); p->d_tape = nullptr; p->tape_bytes = 0; }
            HIP_TRY(hipMalloc((void**)&p->d_tape, need_tape));
            p->tape_bytes = need_tape;
        }
        if (!p->d_grad2_params) HIP_TRY(hipMalloc((void**)&p->d_grad2_params, sizeof(Grad2Params) * 64));
        double* o_lnl = nsplit > 1 ? p->d_part : d_lnl;
        double* o_sum = nsplit > 1 ? p->d_part + items_all : d_sum_dlogt;
        double* o_dex = nsplit > 1 ? p->d_part + 2 * items_all : d_dexch;
        double* o_dlt = d_dlogt ? (nsplit > 1 ? p->d_part + 8 * items_all : d_dlogt) : nullptr;
        double* o_d2 = d_d2logt ? (nsplit > 1 ? p->d_part + (8 + (d_dlogt ? nn : 0)) * items_all : d_d2logt) : nullptr;
        int slot = 0;
        for (int64_t done = 0; done < ncand; done += chunk, ++slot) {
            const int64_t n = std::min<int64_t>(ncand - done, chunk);
            HIP_TRY(launch_lik_eigen_kernel(st, p->d_models.p, d_cand_locus + done, d_cand_exch + done * 6, n, d_eig));
            HIP_TRY(launch_lik_pmat_kernel(st, d_eig, d_blen_vecs, d_cand_vec + done, d_cand_scale + done, d_cand_pidx + done,
                                           d_cand_pfac + done, n, p->nnodes, d_pmat));
            HIP_TRY(launch_grad2_ef_kernel(st, d_eig, d_blen_vecs, d_cand_vec + done, d_cand_scale + done, d_cand_pidx + done,
                                           d_cand_pfac + done, n, p->nnodes, d_ef));
            Grad2Params G2;
            G2.packed = p->d_value_packed; G2.ncols_total = p->ncols; G2.locus_offsets = p->d_offsets.p; G2.col_weight = p->d_col_weight;
            G2.models = p->d_models.p; G2.fops = p->d_grad2_fops.p; G2.rops = p->d_grad2_rops.p; G2.nrops = p->grad2_nrops;
            G2.ntaxa = p->ntaxa; G2.nnodes = p->nnodes; G2.nwords = p->nwords; G2.ntape = p->grad2_ntape; G2.rdepth = p->grad2_rdepth;
            G2.tip_node = p->d_value_tip_node.p; G2.cand_locus = d_cand_locus + done; G2.eig = d_eig; G2.pmat = d_pmat; G2.ef = d_ef;
            G2.ncand = n; G2.nsplit = nsplit; G2.tape = p->d_tape;
            const size_t o = (size_t)done * nsplit;
            G2.out_lnl = o_lnl + o; G2.out_sum_dlogt = o_sum + o; G2.out_dexch = o_dex + o * 6;
            G2.out_dlogt = o_dlt ? o_dlt + o * nn : nullptr; G2.out_d2logt = o_d2 ? o_d2 + o * nn : nullptr;
            if (slot >= 64) { HIP_TRY(hipStreamSynchronize(st)); slot = 0; }   // the parameter blocks in flight are a ring of 64
            Grad2Params* d_par = (Grad2Params*)p->d_grad2_params + slot;
            HIP_TRY(hipMemcpyAsync(d_par, &G2, sizeof(Grad2Params), hipMemcpyHostToDevice, st));
            const int64_t grid = std::min<int64_t>(n * nsplit, grid_max);
            HIP_TRY(launch_locus_grad2_kernel(p->prog.stack_depth, dim3((unsigned)grid), p->grad2_lds, st, d_par));
        }
        if (nsplit > 1) {
            (void)launch_split_sum_kernel(st, o_lnl, d_lnl, ncand, nsplit, 1);
            (void)launch_split_sum_kernel(st, o_sum, d_sum_dlogt, ncand, nsplit, 1);
            (void)launch_split_sum_kernel(st, o_dex, d_dexch, ncand, nsplit, 6);
            if (d_dlogt) (void)launch_split_sum_kernel(st, o_dlt, d_dlogt, ncand, nsplit, (int)nn);
            if (d_d2logt) (void)launch_split_sum_kernel(st, o_d2, d_d2logt, ncand, nsplit, (int)nn);
            HIP_TRY(hipGetLastError());
        }
        return TPHIP_OK;
    }
    GradParams G;
    LikParams& L = G.L;
    L.states = d_states; L.ncols_total = p->ncols; L.locus_offsets = p->d_offsets.p; L.models = p->d_models.p;
    L.col_weight = p->d_col_weight;
    L.lops = p->d_lik_ops.p; L.nops = (int32_t)p->prog.ops.size(); L.nnodes = p->nnodes; L.ntaxa = p->ntaxa;
    L.stack_depth = p->prog.stack_depth; L.cand_locus = d_cand_locus; L.cand_exch = d_cand_exch;
    L.blen_vecs = d_blen_vecs; L.cand_vec = d_cand_vec; L.cand_scale = d_cand_scale; L.cand_pidx = d_cand_pidx;
    L.cand_pfac = d_cand_pfac; L.out = d_lnl;
    G.ntape = p->prog.ntape; G.ncand = ncand; G.nslots = p->grad_slots;
    G.out_dexch = d_dexch; G.out_dlogt = d_dlogt; G.out_sum_dlogt = d_sum_dlogt; G.out_d2logt = d_d2logt;
    if (!p->grad_ok) return fail(TPHIP_ERR_INVALID, "tree too large for the locus-gradient kernel's LDS tables");
    const size_t lds = p->grad_lds;
    L.stage_states = p->grad_stage;
    const int blocks_per_cu = p->grad_blocks_per_cu;   // resident workgroups loop over the candidates; each owns one tape
    const int nsplit = lik_nsplit(p, ncand, kGradBlock);
    L.nsplit = nsplit;
    const size_t nn = (size_t)p->nnodes, items = (size_t)ncand * nsplit;
    if (nsplit > 1) {   // partials: lnl[items], sum[items], dexch[items][6], dlogt[items][nn]
        int rc = grow_part(p, items * (8 + (d_dlogt ? nn : 0) + (d_d2logt ? nn : 0)) * sizeof(double));
        if (rc) return rc;
        L.out = p->d_part;
        G.out_sum_dlogt = p->d_part + items;
        G.out_dexch = p->d_part + 2 * items;
        G.out_dlogt = d_dlogt ? p->d_part + 8 * items : nullptr;
        G.out_d2logt = d_d2logt ? p->d_part + (8 + (d_dlogt ? nn : 0)) * items : nullptr;
    }
    const int64_t grid = std::min<int64_t>((int64_t)items, (int64_t)p->num_cus * blocks_per_cu);
    const size_t need = (size_t)grid * (size_t)std::max(1, p->prog.ntape + p->prog.stack_depth) * 4 * kGradBlock * sizeof(double);
    if (need > p->tape_bytes) {
        if (p->d_tape) { HIP_TRY(hipFree(p->d_tape)); p->d_tape = nullptr; p->tape_bytes = 0; }
        HIP_TRY(hipMalloc((void**)&p->d_tape, need));
        p->tape_bytes = need;
    }
    G.tape = p->d_tape;
    // eigen-systems of the candidates by one thread each (locus_value_launch.hip) instead of thread 0 of every work item
    G.cand_eig = nullptr;
    {
        const size_t need_eig = (size_t)ncand * 36 * sizeof(double);
        if (need_eig > p->grad_eig_bytes) {
            if (p->d_grad_eig) { HIP_TRY(hipFree(p->d_grad_eig)); p->d_grad_eig = nullptr; p->grad_eig_bytes = 0; }
            HIP_TRY(hipMalloc((void**)&p->d_grad_eig, need_eig));
            p->grad_eig_bytes = need_eig;
        }
        HIP_TRY(launch_lik_eigen_kernel((hipStream_t)stream, p->d_models.p, d_cand_locus, d_cand_exch, ncand, p->d_grad_eig));
        G.cand_eig = p->d_grad_eig;
    }
    if (!p->d_grad_params) HIP_TRY(hipMalloc((void**)&p->d_grad_params, sizeof(GradParams)));
    HIP_TRY(hipMemcpyAsync(p->d_grad_params, &G, sizeof(GradParams), hipMemcpyHostToDevice, (hipStream_t)stream));
    HIP_TRY(launch_locus_grad_kernel(dim3((unsigned)grid), lds, (hipStream_t)stream, (const GradParams*)p->d_grad_params));
    if (nsplit > 1) {
        auto sum = [&](const double* part, double* out, int width) {
            (void)launch_split_sum_kernel((hipStream_t)stream, part, out, ncand, nsplit, width);
        };
        sum(L.out, d_lnl, 1);
        sum(G.out_sum_dlogt, d_sum_dlogt, 1);
        sum(G.out_dexch, d_dexch, 6);
        if (d_dlogt) sum(G.out_dlogt, d_dlogt, (int)nn);
        if (d_d2logt) sum(G.out_d2logt, d_d2logt, (int)nn);
        HIP_TRY(hipGetLastError());
    }
    return TPHIP_OK;
}

int tphip_state_histogram_dev(int32_t device, const uint8_t* d_states, int64_t ncols_total, int32_t ntaxa,
                              const int64_t* d_locus_offsets, int64_t nloci, int64_t* d_hist, void* stream) {
    if (tphip_device_count() <= 0) return fail(TPHIP_ERR_NO_DEVICE, "no HIP device visible: libtphip has no CPU path");
    if (!d_states || !d_locus_offsets || !d_hist || nloci < 1 || ntaxa < 1) return fail(TPHIP_ERR_INVALID, "bad arguments");
    HIP_TRY(hipSetDevice(device));
    state_histogram_kernel<<<dim3((unsigned)nloci), dim3(256), 0, (hipStream_t)stream>>>(
        d_states, ncols_total, ntaxa, d_locus_offsets, (unsigned long long*)d_hist);
    HIP_TRY(hipGetLastError());
    return TPHIP_OK;
}

// ---- host-pointer wrappers -------------------------------------------------------------------------

// Host-pointer calls keep their device buffers in the plan (grown on demand, freed with the plan): a hipMalloc /
// hipFree pair per buffer per call cost more than the kernels of a small batch.  Copies are asynchronous on a second
// stream where the caller's memory allows it (pinned: tphip_host_alloc, or registered by the caller): the per-column
// outputs leave the device while the PI kernels run, and rate / flag / nres leave as soon as the site-rate stage ends.
// Pageable memory is copied by the runtime's own staged hipMemcpy (same code path, no overlap).
namespace {
struct HostBuffers {
    uint8_t* states = nullptr; size_t states_bytes = 0;
    char* percol = nullptr; size_t percol_bytes = 0;    // rate | subst | lnl | nres | flag, one allocation
    double* tables = nullptr; size_t tables_bytes = 0;
    void* ws = nullptr; size_t ws_bytes = 0;
    hipStream_t copy_stream = nullptr, run_stream = nullptr;
    hipEvent_t ev_in = nullptr, ev_site = nullptr;
};

int grow(void** ptr, size_t& have, size_t need) {
    if (need <= have && *ptr) return TPHIP_OK;
    if (*ptr) { HIP_TRY(hipFree(*ptr)); *ptr = nullptr; have = 0; }
    HIP_TRY(hipMalloc(ptr, need ? need : 1));
    have = need;
    return TPHIP_OK;
}

bool is_pinned(const void* q) {   // host memory the DMA engines can reach directly
    hipPointerAttribute_t a;
    if (hipPointerGetAttributes(&a, q) != hipSuccess) { (void)hipGetLastError(); return false; }
    return a.type == hipMemoryTypeHost;
}
}  // namespace

struct tphip_host_buffers : HostBuffers {};

static void free_host_buffers(tphip_plan* p) {
    HostBuffers* B = p->hostbuf;
    if (!B) return;
    for (void* q : {(void*)B->states, (void*)B->percol, (void*)B->tables, B->ws}) if (q) (void)hipFree(q);
    if (B->copy_stream) (void)hipStreamDestroy(B->copy_stream);
    if (B->run_stream) (void)hipStreamDestroy(B->run_stream);
    if (B->ev_in) (void)hipEventDestroy(B->ev_in);
    if (B->ev_site) (void)hipEventDestroy(B->ev_site);
    delete B;
    p->hostbuf = nullptr;
}

// Queues one host-pointer pass on the plan's own streams (no wait).  `spitch` = bytes between taxon rows of `states` at the
// caller (= the plan's column count unless the plan is a locus group of a bigger batch); `after` = an event the upload must
// wait for (the previous group's upload: the DMA engine serves one upload after the other, not all of them slowly).
static int host_enqueue(tphip_plan* p, const uint8_t* states, size_t spitch, hipEvent_t after, const double* rates_in,
                        const int32_t* nres_in, double* rate, double* subst, double* lnl, uint8_t* flag, int32_t* nres,
                        double* tables, bool do_site, bool do_pi) {
    HIP_TRY(hipSetDevice(p->device));
    if (!p->hostbuf) {
        p->hostbuf = new tphip_host_buffers();
        HIP_TRY(hipStreamCreateWithFlags(&p->hostbuf->copy_stream, hipStreamNonBlocking));
        HIP_TRY(hipStreamCreateWithFlags(&p->hostbuf->run_stream, hipStreamNonBlocking));
        HIP_TRY(hipEventCreateWithFlags(&p->hostbuf->ev_in, hipEventDisableTiming));
        HIP_TRY(hipEventCreateWithFlags(&p->hostbuf->ev_site, hipEventDisableTiming));
    }
    HostBuffers& B = *p->hostbuf;
    const size_t n = (size_t)p->ncols, W = (size_t)tphip_plan_table_width(p);
    // per-column block: rate | subst | lnl (8 B each) | nres (4 B) | flag (1 B), each part 256-byte aligned
    const size_t o_rate = 0, o_subst = align_up(8 * n, 256), o_lnl = o_subst + align_up(8 * n, 256),
                 o_nres = o_lnl + align_up(8 * n, 256), o_flag = o_nres + align_up(4 * n, 256);
    int rc = grow((void**)&B.percol, B.percol_bytes, o_flag + align_up(n, 256));
    if (rc) return rc;
    rc = grow(&B.ws, B.ws_bytes, p->ws_total);
    if (rc) return rc;
    double* d_rate = (double*)(B.percol + o_rate);
    double* d_subst = (double*)(B.percol + o_subst);
    double* d_lnl = (double*)(B.percol + o_lnl);
    int32_t* d_nres = (int32_t*)(B.percol + o_nres);
    uint8_t* d_flag = (uint8_t*)(B.percol + o_flag);
    hipStream_t cs = B.copy_stream, rs = B.run_stream;
    if (do_site) {
        if (!states || !rate || !subst || !lnl || !flag || !nres) return fail(TPHIP_ERR_INVALID, "null host pointer");
        rc = grow((void**)&B.states, B.states_bytes, n * (size_t)p->ntaxa + 1);
        if (rc) return rc;
        if (after) HIP_TRY(hipStreamWaitEvent(rs, after, 0));
        if (spitch == n) HIP_TRY(hipMemcpyAsync(B.states, states, n * (size_t)p->ntaxa, hipMemcpyHostToDevice, rs));
        else HIP_TRY(hipMemcpy2DAsync(B.states, n, states, spitch, n, (size_t)p->ntaxa, hipMemcpyHostToDevice, rs));
        HIP_TRY(hipEventRecord(B.ev_in, rs));
        rc = launch_site_rates(p, B.states, d_rate, d_subst, d_lnl, d_flag, d_nres, B.ws, rs, -1);
        if (rc) return rc;
        HIP_TRY(hipEventRecord(B.ev_site, rs));
    } else {
        if (!rates_in) return fail(TPHIP_ERR_INVALID, "null host pointer");
        HIP_TRY(hipMemcpyAsync(d_rate, rates_in, sizeof(double) * n, hipMemcpyHostToDevice, rs));
        if (nres_in) HIP_TRY(hipMemcpyAsync(d_nres, nres_in, sizeof(int32_t) * n, hipMemcpyHostToDevice, rs));
    }
    if (do_pi) {
        if (!tables) return fail(TPHIP_ERR_INVALID, "null host pointer");
        rc = grow((void**)&B.tables, B.tables_bytes, sizeof(double) * W * (size_t)p->nloci + 8);
        if (rc) return rc;
        rc = launch_pi_tables(p, d_rate, (do_site || nres_in) ? d_nres : nullptr, B.tables, B.ws, rs, -1);
        if (rc) return rc;
    }
    if (do_site) {
        // the per-column results go home on the copy stream while the PI kernels (which only read them) run; with
        // pageable destinations the runtime stages the copies itself and they simply run in order
        const bool overlap = is_pinned(rate) && is_pinned(subst) && is_pinned(lnl) && is_pinned(flag) && is_pinned(nres);
        hipStream_t os = overlap ? cs : rs;
        if (overlap) HIP_TRY(hipStreamWaitEvent(cs, B.ev_site, 0));
        HIP_TRY(hipMemcpyAsync(rate, d_rate, sizeof(double) * n, hipMemcpyDeviceToHost, os));
        HIP_TRY(hipMemcpyAsync(subst, d_subst, sizeof(double) * n, hipMemcpyDeviceToHost, os));
        HIP_TRY(hipMemcpyAsync(lnl, d_lnl, sizeof(double) * n, hipMemcpyDeviceToHost, os));
        HIP_TRY(hipMemcpyAsync(flag, d_flag, n, hipMemcpyDeviceToHost, os));
        HIP_TRY(hipMemcpyAsync(nres, d_nres, sizeof(int32_t) * n, hipMemcpyDeviceToHost, os));
    }
    if (do_pi) HIP_TRY(hipMemcpyAsync(tables, B.tables, sizeof(double) * W * (size_t)p->nloci, hipMemcpyDeviceToHost, rs));
    return TPHIP_OK;
}

static int host_wait(tphip_plan* p) {
    if (!p->hostbuf) return TPHIP_OK;
    HIP_TRY(hipStreamSynchronize(p->hostbuf->run_stream));
    HIP_TRY(hipStreamSynchronize(p->hostbuf->copy_stream));
    return TPHIP_OK;
}

constexpr int64_t kHostSplitMinColumns = (int64_t)1 << 21;   // below this a pass is too short for the pipeline to pay

static int host_run(tphip_plan* p, const uint8_t* states, const double* rates_in, const int32_t* nres_in, double* rate,
                    double* subst, double* lnl, uint8_t* flag, int32_t* nres, double* tables, bool do_site, bool do_pi,
                    size_t row_pitch = 0) {
    if (!p) return fail(TPHIP_ERR_INVALID, "null plan");
    const size_t pitch = row_pitch ? row_pitch : (size_t)p->ncols;   // bytes between taxon rows of `states` at the caller
    if (pitch < (size_t)p->ncols) return fail(TPHIP_ERR_INVALID, "row_pitch smaller than the plan's column count");
    // Big batches from pinned host memory: locus groups in a pipeline -- the upload of group k + 1 (and the download of group
    // k - 1) run while group k computes.  Results do not depend on how a batch is cut (per-column results depend on the
    // column and its locus' model only, a locus' PI row on its own columns in a fixed order).
    const bool split = do_site && !p->is_part && p->saved && p->host_split > 1 && p->nloci >= 2 * p->host_split &&
                       p->ncols >= kHostSplitMinColumns && states && rate && subst && lnl && flag && nres &&
                       is_pinned(states) && is_pinned(rate) && is_pinned(subst) && is_pinned(lnl) && is_pinned(flag) &&
                       is_pinned(nres) && (!do_pi || (tables && is_pinned(tables)));
    if (!split) {
        // always drain: a failed enqueue may already have copies in flight that touch the caller's buffers
        const int rc = host_enqueue(p, states, pitch, nullptr, rates_in, nres_in, rate, subst, lnl, flag, nres, tables,
                                    do_site, do_pi);
        const int rw = host_wait(p);
        return rc ? rc : rw;
    }
    int rc = make_parts(p);
    if (rc) return rc;
    if (p->parts.empty()) {   // the batch does not cut into groups with columns
        rc = host_enqueue(p, states, pitch, nullptr, rates_in, nres_in, rate, subst, lnl, flag, nres, tables, do_site, do_pi);
        const int rw = host_wait(p);
        return rc ? rc : rw;
    }
    const size_t W = (size_t)tphip_plan_table_width(p);
    p->last_run_in_parts = true;
    hipEvent_t after = nullptr;
    for (size_t k = 0; k < p->parts.size(); ++k) {
        tphip_plan* q = p->parts[k];
        const int64_t l0 = p->part_locus[k], c0 = p->h_offsets[l0];
        rc = host_enqueue(q, states + c0, pitch, after, nullptr, nullptr, rate + c0, subst + c0, lnl + c0, flag + c0,
                          nres + c0, do_pi ? tables + (size_t)l0 * W : nullptr, true, do_pi);
        if (rc) break;
        after = q->hostbuf->ev_in;
    }
    for (tphip_plan* q : p->parts) {
        const int rw = host_wait(q);
        if (!rc) rc = rw;
    }
    return rc;
}

void* tphip_host_alloc(size_t bytes) {
    void* q = nullptr;
    if (tphip_device_count() <= 0) { fail(TPHIP_ERR_NO_DEVICE, "no HIP device visible: libtphip has no CPU path"); return nullptr; }
    if (hipHostMalloc(&q, bytes ? bytes : 1, hipHostMallocDefault) != hipSuccess) { fail(TPHIP_ERR_HIP, "hipHostMalloc failed"); return nullptr; }
    return q;
}

int tphip_host_free(void* ptr) {
    if (ptr) HIP_TRY(hipHostFree(ptr));
    return TPHIP_OK;
}

int tphip_site_rates(tphip_plan* p, const uint8_t* states, double* rate, double* subst, double* lnl, uint8_t* flag,
                     int32_t* nres) {
    return host_run(p, states, nullptr, nullptr, rate, subst, lnl, flag, nres, nullptr, true, false);
}

int tphip_pi_tables(tphip_plan* p, const double* rates, const int32_t* nres, double* tables) {
    return host_run(p, nullptr, rates, nres, nullptr, nullptr, nullptr, nullptr, nullptr, tables, false, true);
}

int tphip_run_fused(tphip_plan* p, const uint8_t* states, double* rate, double* subst, double* lnl, uint8_t* flag,
                    int32_t* nres, double* tables) {
    return host_run(p, states, nullptr, nullptr, rate, subst, lnl, flag, nres, tables, true, true);
}

int tphip_run_fused_pitched(tphip_plan* p, const uint8_t* states, int64_t row_pitch, double* rate, double* subst, double* lnl,
                            uint8_t* flag, int32_t* nres, double* tables) {
    if (row_pitch < 0) return fail(TPHIP_ERR_INVALID, "negative row_pitch");
    return host_run(p, states, nullptr, nullptr, rate, subst, lnl, flag, nres, tables, true, true, (size_t)row_pitch);
}

int tphip_townsend_pi_dense(int32_t device, const double* rates, int64_t n, const double* times, int32_t n_times, double* out) {
    if (tphip_device_count() <= 0) return fail(TPHIP_ERR_NO_DEVICE, "no HIP device visible: libtphip has no CPU path");
    if (n < 0 || n_times < 0) return fail(TPHIP_ERR_INVALID, "bad arguments");
    if (n == 0 || n_times == 0) return TPHIP_OK;
    if (!rates || !times || !out) return fail(TPHIP_ERR_INVALID, "null host pointer");
    HIP_TRY(hipSetDevice(device));
    Scratch S;
    double* d_r = S.get<double>((size_t)n);
    double* d_t = S.get<double>((size_t)n_times);
    double* d_o = S.get<double>((size_t)n * (size_t)n_times);
    if (!d_r || !d_t || !d_o) return fail(TPHIP_ERR_HIP, "hipMalloc failed for the dense PI matrix");
    HIP_TRY(hipMemcpy(d_r, rates, sizeof(double) * (size_t)n, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(d_t, times, sizeof(double) * (size_t)n_times, hipMemcpyHostToDevice));
    int rc = tphip_townsend_pi_dense_dev(device, d_r, n, d_t, n_times, d_o, nullptr);
    if (rc) return rc;
    HIP_TRY(hipMemcpy(out, d_o, sizeof(double) * (size_t)n * (size_t)n_times, hipMemcpyDeviceToHost));
    return TPHIP_OK;
}

int tphip_quad_townsend(int32_t device, const double* rates, int64_t n, double a, double b, int32_t integ_mode,
                        double* integral, double* abserr) {
    if (tphip_device_count() <= 0) return fail(TPHIP_ERR_NO_DEVICE, "no HIP device visible: libtphip has no CPU path");
    if (n < 0) return fail(TPHIP_ERR_INVALID, "bad arguments");
    if (n == 0) return TPHIP_OK;
    if (!rates || !integral || !abserr) return fail(TPHIP_ERR_INVALID, "null host pointer");
    HIP_TRY(hipSetDevice(device));
    Scratch S;
    double* d_r = S.get<double>((size_t)n);
    double* d_i = S.get<double>((size_t)n);
    double* d_e = S.get<double>((size_t)n);
    if (!d_r || !d_i || !d_e) return fail(TPHIP_ERR_HIP, "hipMalloc failed");
    HIP_TRY(hipMemcpy(d_r, rates, sizeof(double) * (size_t)n, hipMemcpyHostToDevice));
    int rc = tphip_quad_townsend_dev(device, d_r, n, a, b, integ_mode, d_i, d_e, nullptr);
    if (rc) return rc;
    HIP_TRY(hipMemcpy(integral, d_i, sizeof(double) * (size_t)n, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(abserr, d_e, sizeof(double) * (size_t)n, hipMemcpyDeviceToHost));
    return TPHIP_OK;
}

int tphip_eval_columns_dev(tphip_plan* p, const uint8_t* d_s, const double* d_u, double* d_f, double* d_g, double* d_h,
                           void* stream) {
    if (!p || !d_s || !d_u || !d_f || !d_g || !d_h) return fail(TPHIP_ERR_INVALID, "null argument");
    HIP_TRY(hipSetDevice(p->device));
    EvalParams E;
    E.S.states = d_s; E.S.ncols_total = p->ncols; E.S.models = p->d_models.p; E.S.ops = p->d_ops.p;
    E.S.nops = (int32_t)p->prog.ops.size(); E.S.stack_depth = p->prog.stack_depth; E.S.chrono_length = p->prog.chrono_length;
    E.S.locus_offsets = p->d_offsets.p; E.S.chunk_locus = p->d_site_chunk_locus.p; E.S.chunk_index = p->d_site_chunk_index.p;
    E.S.chunk_cols = p->site_chunk_cols;
    E.S.packed = nullptr; E.S.nwords = 0;
    E.S.work_cols = nullptr; E.S.work_cols2 = nullptr; E.S.work_count = nullptr; E.S.work_prefix = nullptr; E.S.nloci = p->nloci; E.S.persistent = 0; E.S.first_round = 0; E.S.mixed_few_waves = 0; E.S.mixed_switch_cols = 0; E.S.first_fraction = 1.0; E.S.ncat = p->ncat; E.S.cat = p->d_cat.p; E.S.rate = nullptr; E.S.subst = nullptr; E.S.lnl = nullptr;
    E.S.flag = nullptr; E.S.eval_counter = nullptr; E.S.spill = nullptr; E.S.lds_depth = p->prog.stack_depth;
    E.u = d_u; E.f = d_f; E.g = d_g; E.h = d_h;
    const size_t lds = (kSiteLdsHeader + (size_t)p->prog.stack_depth * 12 * kSiteBlock) * sizeof(double);
    if (p->n_site_chunks > 0) HIP_TRY(launch_eval_columns_kernel(dim3((unsigned)p->n_site_chunks), lds, (hipStream_t)stream, E));
    return TPHIP_OK;
}

int tphip_eval_columns(tphip_plan* p, const uint8_t* states, const double* u, double* f, double* g, double* h) {
    if (!p || !states || !u || !f || !g || !h) return fail(TPHIP_ERR_INVALID, "null argument");
    HIP_TRY(hipSetDevice(p->device));
    Scratch S;
    const size_t n = (size_t)p->ncols;
    uint8_t* d_s = S.get<uint8_t>(n * (size_t)p->ntaxa);
    double* d_u = S.get<double>(n);
    double* d_f = S.get<double>(n);
    double* d_g = S.get<double>(n);
    double* d_h = S.get<double>(n);
    if (!d_s || !d_u || !d_f || !d_g || !d_h) return fail(TPHIP_ERR_HIP, "hipMalloc failed");
    HIP_TRY(hipMemcpy(d_s, states, n * (size_t)p->ntaxa, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(d_u, u, sizeof(double) * n, hipMemcpyHostToDevice));
    int rc = tphip_eval_columns_dev(p, d_s, d_u, d_f, d_g, d_h, nullptr);
    if (rc) return rc;
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(f, d_f, sizeof(double) * n, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(g, d_g, sizeof(double) * n, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(h, d_h, sizeof(double) * n, hipMemcpyDeviceToHost));
    return TPHIP_OK;
}

int tphip_plan_set_column_weights(tphip_plan* p, const double* weights) {
    if (!p) return fail(TPHIP_ERR_INVALID, "null plan");
    HIP_TRY(hipSetDevice(p->device));
    if (p->d_col_weight) { HIP_TRY(hipFree(p->d_col_weight)); p->d_col_weight = nullptr; }
    if (!weights || p->ncols == 0) return TPHIP_OK;
    for (int64_t c = 0; c < p->ncols; ++c)
        if (!(weights[c] >= 0.0)) return fail(TPHIP_ERR_INVALID, "column weights must be >= 0");
    HIP_TRY(hipMalloc((void**)&p->d_col_weight, sizeof(double) * (size_t)p->ncols));
    HIP_TRY(hipMemcpy(p->d_col_weight, weights, sizeof(double) * (size_t)p->ncols, hipMemcpyHostToDevice));
    return TPHIP_OK;
}

// per-locus eigen-systems from DEVICE arrays pi [L][4] (floored, any scale) and exch [L][6]; for stage1_driver.hip
int tphip_internal_set_models_dev(tphip_plan* p, const double* d_pi, const double* d_exch, void* stream) {
    const int bs = 64;
    gtr_setup_kernel<<<dim3((unsigned)((p->nloci + bs - 1) / bs)), dim3(bs), 0, (hipStream_t)stream>>>(d_pi, d_exch, p->nloci, p->d_models.p);
    HIP_TRY(hipGetLastError());
    return TPHIP_OK;
}

// the plan's record of its base frequencies after they were replaced on the device (stage1_driver.hip: empirical_pi)
int tphip_internal_store_pi(tphip_plan* p, const double* pi) {
    if (!p || !p->saved) return fail(TPHIP_ERR_INVALID, "plan has no saved descriptor");
    p->saved->pi.assign(pi, pi + 4 * (size_t)p->nloci);
    p->saved->d.pi = p->saved->pi.data();
    for (tphip_plan* q : p->parts) (void)tphip_plan_destroy(q);
    p->parts.clear();
    return TPHIP_OK;
}

int tphip_plan_set_models(tphip_plan* p, const double* pi, const double* exch) {
    if (!p) return fail(TPHIP_ERR_INVALID, "null plan");
    if (!p->saved) return fail(TPHIP_ERR_INVALID, "plan has no saved descriptor");
    const size_t L = (size_t)p->nloci;
    if (pi) {
        for (size_t l = 0; l < L; ++l) {
            int npos = 0;
            for (int k = 0; k < 4; ++k) {
                if (!(pi[l * 4 + k] >= 0.0) || !std::isfinite(pi[l * 4 + k]))
                    return fail(TPHIP_ERR_INVALID, "base frequencies of locus " + std::to_string(l) + " must be finite and >= 0");
                npos += pi[l * 4 + k] > 0.0;
            }
            if (npos < 2) return fail(TPHIP_ERR_INVALID, "locus " + std::to_string(l) + " has fewer than two bases with a positive frequency");
        }
        p->saved->pi.assign(pi, pi + 4 * L);
    }
    if (exch) {
        for (size_t i = 0; i < 6 * L; ++i)
            if (!(exch[i] >= 0.0)) return fail(TPHIP_ERR_INVALID, "exchangeabilities must be >= 0");
        p->saved->exch.assign(exch, exch + 6 * L);
    }
    p->saved->d.pi = p->saved->pi.data();
    p->saved->d.exch = p->saved->exch.data();
    HIP_TRY(hipSetDevice(p->device));
    std::vector<double> hpi(p->saved->pi);
    for (size_t l = 0; l < L; ++l) {   // same floor as at plan creation
        double sum = 0;
        for (int k = 0; k < 4; ++k) sum += hpi[4 * l + k];
        for (int k = 0; k < 4; ++k) hpi[4 * l + k] = std::max(hpi[4 * l + k], kPiFloor * sum);
    }
    Scratch S;
    double* d_pi = S.get<double>(4 * L);
    double* d_ex = S.get<double>(6 * L);
    if (!d_pi || !d_ex) return fail(TPHIP_ERR_HIP, "hipMalloc failed");
    HIP_TRY(hipMemcpy(d_pi, hpi.data(), sizeof(double) * 4 * L, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(d_ex, p->saved->exch.data(), sizeof(double) * 6 * L, hipMemcpyHostToDevice));
    int rc = tphip_internal_set_models_dev(p, d_pi, d_ex, nullptr);
    if (rc) return rc;
    HIP_TRY(hipDeviceSynchronize());
    // locus groups of the pipelined host path were built from the old models: rebuilt on next use
    for (tphip_plan* q : p->parts) (void)tphip_plan_destroy(q);
    p->parts.clear();
    return TPHIP_OK;
}

int tphip_free_device(tphip_plan* p, void* d_ptr) {
    if (!p) return fail(TPHIP_ERR_INVALID, "null plan");
    HIP_TRY(hipSetDevice(p->device));
    if (d_ptr && d_ptr == (void*)p->lib_states) {   // the alignment cache goes: so do the codes packed from it
        if (p->d_value_packed) { HIP_TRY(hipFree(p->d_value_packed)); p->d_value_packed = nullptr; }
        p->lib_states = nullptr;
    }
    if (d_ptr) HIP_TRY(hipFree(d_ptr));
    return TPHIP_OK;
}

// Candidate batch staged through the plan's arena: one pinned buffer, one H2D copy in, one D2H copy out.
namespace {
struct CandBatch {
    size_t off_locus, off_exch, off_blen, off_vec, off_scale, off_pidx, off_pfac, in_bytes;
    size_t off_out, out_bytes, total;
};

int arena_reserve(tphip_plan* p, size_t bytes) {
    if (bytes <= p->arena_bytes) return TPHIP_OK;
    const size_t want = std::max(bytes, p->arena_bytes * 2);
    if (p->d_arena) { HIP_TRY(hipFree(p->d_arena)); p->d_arena = nullptr; }
    if (p->h_arena) { HIP_TRY(hipHostFree(p->h_arena)); p->h_arena = nullptr; }
    p->arena_bytes = 0;
    HIP_TRY(hipMalloc((void**)&p->d_arena, want));
    HIP_TRY(hipHostMalloc((void**)&p->h_arena, want, hipHostMallocDefault));
    p->arena_bytes = want;
    return TPHIP_OK;
}

int check_candidates(const tphip_plan* p, int64_t nvec, int64_t ncand, const int32_t* cand_locus, const int32_t* cand_vec,
                     const int32_t* cand_pidx) {
    for (int64_t c = 0; c < ncand; ++c) {
        if (cand_locus[c] < 0 || cand_locus[c] >= p->nloci) return fail(TPHIP_ERR_INVALID, "cand_locus out of range");
        if (cand_vec[c] < 0 || cand_vec[c] >= nvec) return fail(TPHIP_ERR_INVALID, "cand_vec out of range");
        if (cand_pidx[c] >= p->nnodes) return fail(TPHIP_ERR_INVALID, "cand_pidx out of range");
    }
    return TPHIP_OK;
}

// device copy of the alignment: the caller's cache, or a scratch buffer for this call
int stage_alignment(tphip_plan* p, const uint8_t* states, void** d_states_cache, Scratch& S, uint8_t** d_s) {
    const size_t nb = (size_t)p->ncols * (size_t)p->ntaxa;
    *d_s = d_states_cache ? (uint8_t*)*d_states_cache : nullptr;
    if (*d_s) return TPHIP_OK;
    if (d_states_cache) {
        HIP_TRY(hipMalloc((void**)d_s, nb ? nb : 1));
        *d_states_cache = *d_s;
    } else {
        *d_s = S.get<uint8_t>(nb);
        if (!*d_s) return fail(TPHIP_ERR_HIP, "hipMalloc failed");
    }
    HIP_TRY(hipMemcpy(*d_s, states, nb, hipMemcpyHostToDevice));
    if (d_states_cache && p->value_cols > 0 && p->ncols > 0) {
        // a copy that stays: pack its state codes once for locus_value_kernel (8 per word, tip order) instead of once per
        // likelihood evaluation.  Only for the copy made here -- the library cannot know when a caller's own device array changes.
        if (p->d_value_packed) { HIP_TRY(hipFree(p->d_value_packed)); p->d_value_packed = nullptr; }
        p->lib_states = nullptr;
        HIP_TRY(hipMalloc((void**)&p->d_value_packed, sizeof(uint32_t) * (size_t)p->nwords * (size_t)p->ncols));
        HIP_TRY(launch_value_pack_codes_kernel(nullptr, *d_s, p->ncols, p->d_tip_taxon.p, p->nwords, p->d_value_packed));
        HIP_TRY(hipStreamSynchronize(nullptr));   // later launches may come on any stream
        p->lib_states = *d_s;
    }
    return TPHIP_OK;
}

}  // namespace

// device copy of the alignment in the caller's cache slot, for the other translation units (stage1_driver.hip)
int tphip_internal_stage_alignment(tphip_plan* p, const uint8_t* states, void** d_states_cache, uint8_t** d_s) {
    Scratch S;
    return stage_alignment(p, states, d_states_cache, S, d_s);
}

namespace {
int stage_candidates(tphip_plan* p, int64_t nvec, const double* blen_vecs, int64_t ncand, const int32_t* cand_locus,
                     const double* cand_exch, const int32_t* cand_vec, const double* cand_scale, const int32_t* cand_pidx,
                     const double* cand_pfac, size_t out_doubles, CandBatch* B) {
    const size_t n = (size_t)ncand, nn = (size_t)p->nnodes;
    size_t off = 0;
    auto take = [&](size_t bytes) { const size_t o = off; off = align_up(off + bytes, 256); return o; };
    B->off_locus = take(sizeof(int32_t) * n);
    B->off_exch = take(sizeof(double) * n * 6);
    B->off_blen = take(sizeof(double) * (size_t)nvec * nn);
    B->off_vec = take(sizeof(int32_t) * n);
    B->off_scale = take(sizeof(double) * n);
    B->off_pidx = take(sizeof(int32_t) * n);
    B->off_pfac = take(sizeof(double) * n);
    B->in_bytes = off;
    B->off_out = take(sizeof(double) * out_doubles);
    B->out_bytes = sizeof(double) * out_doubles;
    B->total = off;
    int rc = arena_reserve(p, B->total);
    if (rc) return rc;
    char* h = p->h_arena;
    memcpy(h + B->off_locus, cand_locus, sizeof(int32_t) * n);
    memcpy(h + B->off_exch, cand_exch, sizeof(double) * n * 6);
    memcpy(h + B->off_blen, blen_vecs, sizeof(double) * (size_t)nvec * nn);
    memcpy(h + B->off_vec, cand_vec, sizeof(int32_t) * n);
    memcpy(h + B->off_scale, cand_scale, sizeof(double) * n);
    memcpy(h + B->off_pidx, cand_pidx, sizeof(int32_t) * n);
    memcpy(h + B->off_pfac, cand_pfac, sizeof(double) * n);
    HIP_TRY(hipMemcpyAsync(p->d_arena, h, B->in_bytes, hipMemcpyHostToDevice, nullptr));
    return TPHIP_OK;
}
}  // namespace

int tphip_locus_loglik(tphip_plan* p, const uint8_t* states, void** d_states_cache, int64_t nvec, const double* blen_vecs,
                       int64_t ncand, const int32_t* cand_locus, const double* cand_exch, const int32_t* cand_vec,
                       const double* cand_scale, const int32_t* cand_pidx, const double* cand_pfac, double* out) {
    if (!p || !states || ncand < 0 || nvec < 0 ||
        (ncand && (!blen_vecs || !cand_locus || !cand_exch || !cand_vec || !cand_scale || !cand_pidx || !cand_pfac || !out)))
        return fail(TPHIP_ERR_INVALID, "null argument");
    if (ncand == 0) return TPHIP_OK;
    int rc = check_candidates(p, nvec, ncand, cand_locus, cand_vec, cand_pidx);
    if (rc) return rc;
    HIP_TRY(hipSetDevice(p->device));
    Scratch S;
    uint8_t* d_s = nullptr;
    rc = stage_alignment(p, states, d_states_cache, S, &d_s);
    if (rc) return rc;
    CandBatch B;
    rc = stage_candidates(p, nvec, blen_vecs, ncand, cand_locus, cand_exch, cand_vec, cand_scale, cand_pidx, cand_pfac,
                          (size_t)ncand, &B);
    if (rc) return rc;
    char* d = p->d_arena;
    rc = tphip_locus_loglik_dev(p, d_s, ncand, (const int32_t*)(d + B.off_locus), (const double*)(d + B.off_exch),
                                (const double*)(d + B.off_blen), (const int32_t*)(d + B.off_vec),
                                (const double*)(d + B.off_scale), (const int32_t*)(d + B.off_pidx),
                                (const double*)(d + B.off_pfac), (double*)(d + B.off_out), nullptr);
    if (rc) return rc;
    HIP_TRY(hipMemcpyAsync(p->h_arena + B.off_out, d + B.off_out, B.out_bytes, hipMemcpyDeviceToHost, nullptr));
    HIP_TRY(hipStreamSynchronize(nullptr));
    memcpy(out, p->h_arena + B.off_out, B.out_bytes);
    return TPHIP_OK;
}

int tphip_locus_gradient(tphip_plan* p, const uint8_t* states, void** d_states_cache, int64_t nvec, const double* blen_vecs,
                         int64_t ncand, const int32_t* cand_locus, const double* cand_exch, const int32_t* cand_vec,
                         const double* cand_scale, const int32_t* cand_pidx, const double* cand_pfac, double* lnl,
                         double* dexch, double* dlogt, double* sum_dlogt, double* d2logt) {
    if (!p || !states || ncand < 0 || nvec < 0 ||
        (ncand && (!blen_vecs || !cand_locus || !cand_exch || !cand_vec || !cand_scale || !cand_pidx || !cand_pfac || !lnl ||
                   !dexch || !sum_dlogt)))
        return fail(TPHIP_ERR_INVALID, "null argument");
    if (ncand == 0) return TPHIP_OK;
    int rc = check_candidates(p, nvec, ncand, cand_locus, cand_vec, cand_pidx);
    if (rc) return rc;
    HIP_TRY(hipSetDevice(p->device));
    Scratch S;
    uint8_t* d_s = nullptr;
    rc = stage_alignment(p, states, d_states_cache, S, &d_s);
    if (rc) return rc;
    const size_t n = (size_t)ncand, nn = (size_t)p->nnodes;
    CandBatch B;   // outputs: lnl[n] | sum_dlogt[n] | dexch[6n] | dlogt[n * nn] (optional) | d2logt[n * nn] (optional)
    rc = stage_candidates(p, nvec, blen_vecs, ncand, cand_locus, cand_exch, cand_vec, cand_scale, cand_pidx, cand_pfac,
                          n * (8 + (dlogt ? nn : 0) + (d2logt ? nn : 0)), &B);
    if (rc) return rc;
    char* d = p->d_arena;
    double* d_o = (double*)(d + B.off_out);
    rc = tphip_locus_gradient_dev(p, d_s, ncand, (const int32_t*)(d + B.off_locus), (const double*)(d + B.off_exch),
                                  (const double*)(d + B.off_blen), (const int32_t*)(d + B.off_vec),
                                  (const double*)(d + B.off_scale), (const int32_t*)(d + B.off_pidx),
                                  (const double*)(d + B.off_pfac), d_o, d_o + 2 * n, dlogt ? d_o + 8 * n : nullptr, d_o + n,
                                  d2logt ? d_o + (8 + (dlogt ? nn : 0)) * n : nullptr, nullptr);
    if (rc) return rc;
    HIP_TRY(hipMemcpyAsync(p->h_arena + B.off_out, d + B.off_out, B.out_bytes, hipMemcpyDeviceToHost, nullptr));
    HIP_TRY(hipStreamSynchronize(nullptr));
    const double* h_o = (const double*)(p->h_arena + B.off_out);
    memcpy(lnl, h_o, sizeof(double) * n);
    memcpy(sum_dlogt, h_o + n, sizeof(double) * n);
    memcpy(dexch, h_o + 2 * n, sizeof(double) * n * 6);
    if (dlogt) memcpy(dlogt, h_o + 8 * n, sizeof(double) * n * nn);
    if (d2logt) memcpy(d2logt, h_o + (8 + (dlogt ? nn : 0)) * n, sizeof(double) * n * nn);
    return TPHIP_OK;
}

// Unique site patterns of device-resident columns (see tphip_compress_columns): temporaries in `tmp`, results in `keep`.
static int compress_core(const uint8_t* d_s, int64_t ncols_total, int32_t ntaxa, const int64_t* d_off, int64_t nloci, bool want_map,
                         Scratch& tmp, Scratch& keep, uint8_t** d_out_p, int64_t** d_newoff_p, double** d_w_p, int64_t** d_map_p,
                         int64_t* npat_p) {
    const size_t n = (size_t)ncols_total;
    const int32_t nwords = (ntaxa + 7) / 8;
    uint32_t* d_packed = tmp.get<uint32_t>(n * (size_t)nwords);
    uint64_t* d_key = tmp.get<uint64_t>(n);
    uint64_t* d_key2 = tmp.get<uint64_t>(n);
    uint64_t* d_h2 = tmp.get<uint64_t>(n);
    uint32_t* d_col = tmp.get<uint32_t>(n);
    uint32_t* d_col2 = tmp.get<uint32_t>(n);
    int32_t* d_head = tmp.get<int32_t>(n);
    int64_t* d_incl = tmp.get<int64_t>(n);
    int64_t* d_newoff = keep.get<int64_t>((size_t)nloci + 1);
    int64_t* d_map = want_map ? keep.get<int64_t>(n) : nullptr;
    if (!d_packed || !d_key || !d_key2 || !d_h2 || !d_col || !d_col2 || !d_head || !d_incl || !d_newoff || (want_map && !d_map))
        return fail(TPHIP_ERR_HIP, "hipMalloc failed");
    const unsigned blocks = (unsigned)((n + 255) / 256);
    pack_hash_kernel<<<dim3(blocks), dim3(256)>>>(d_s, ncols_total, ntaxa, nwords, d_off, nloci, d_packed, d_key, d_h2, d_col);
    HIP_TRY(hipGetLastError());
    int locus_bits = 1;
    while (((int64_t)1 << locus_bits) < nloci) ++locus_bits;
    const int end_bit = std::min(64, kPatHashBits + locus_bits);
    size_t tmp_sort = 0, tmp_scan = 0;
    HIP_TRY(hipcub::DeviceRadixSort::SortPairs(nullptr, tmp_sort, d_key, d_key2, d_col, d_col2, (int)n, 0, end_bit));
    HIP_TRY(hipcub::DeviceScan::InclusiveSum(nullptr, tmp_scan, d_head, d_incl, (int)n));
    void* d_tmp = tmp.get<char>(std::max(tmp_sort, tmp_scan));
    if (!d_tmp) return fail(TPHIP_ERR_HIP, "hipMalloc failed");
    HIP_TRY(hipcub::DeviceRadixSort::SortPairs(d_tmp, tmp_sort, d_key, d_key2, d_col, d_col2, (int)n, 0, end_bit));
    head_flag_kernel<<<dim3(blocks), dim3(256)>>>(d_key2, d_col2, d_h2, d_packed, nwords, ncols_total, d_head);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipcub::DeviceScan::InclusiveSum(d_tmp, tmp_scan, d_head, d_incl, (int)n));
    int64_t npat = 0;
    HIP_TRY(hipMemcpy(&npat, d_incl + (n - 1), sizeof(int64_t), hipMemcpyDeviceToHost));
    uint8_t* d_out = keep.get<uint8_t>((size_t)npat * (size_t)ntaxa);
    int32_t* d_count = tmp.get<int32_t>((size_t)npat);
    double* d_w = keep.get<double>((size_t)npat);
    if (!d_out || !d_count || !d_w) return fail(TPHIP_ERR_HIP, "hipMalloc failed");
    HIP_TRY(hipMemset(d_count, 0, sizeof(int32_t) * (size_t)npat));
    pattern_scatter_kernel<<<dim3(blocks), dim3(256)>>>(d_col2, d_head, d_incl, d_packed, nwords, ntaxa, ncols_total, npat, d_out,
                                                      d_count, d_map);
    pattern_offsets_kernel<<<dim3((unsigned)((nloci + 256) / 256)), dim3(256)>>>(d_off, nloci, ncols_total, d_incl, npat, d_newoff);
    count_to_weight_kernel<<<dim3((unsigned)((npat + 255) / 256)), dim3(256)>>>(d_count, npat, d_w);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipDeviceSynchronize());
    *d_out_p = d_out; *d_newoff_p = d_newoff; *d_w_p = d_w; *npat_p = npat;
    if (d_map_p) *d_map_p = d_map;
    return TPHIP_OK;
}

// device-to-device twin for the other translation units (stage1_driver.hip); the three result arrays are the caller's to hipFree
int tphip_internal_compress_dev(const uint8_t* d_s, int64_t ncols_total, int32_t ntaxa, const int64_t* d_off, int64_t nloci,
                                uint8_t** d_out, int64_t** d_newoff, double** d_w, int64_t* npat) {
    if (ncols_total >= ((int64_t)1 << 31)) return fail(TPHIP_ERR_INVALID, "too many columns for one call (2^31)");
    if (nloci >= ((int64_t)1 << (64 - kPatHashBits))) return fail(TPHIP_ERR_INVALID, "too many loci for one call");
    Scratch tmp, keep;
    int rc = compress_core(d_s, ncols_total, ntaxa, d_off, nloci, false, tmp, keep, d_out, d_newoff, d_w, nullptr, npat);
    if (!rc) keep.bufs.clear();   // ownership passes to the caller
    return rc;
}

int tphip_compress_columns(int32_t device, const uint8_t* states, int64_t ncols_total, int32_t ntaxa,
                           const int64_t* locus_offsets, int64_t nloci, uint8_t* out_states, int64_t* out_offsets,
                           double* out_weight, int64_t* out_map, int64_t* out_npatterns) {
    if (tphip_device_count() <= 0) return fail(TPHIP_ERR_NO_DEVICE, "no HIP device visible: libtphip has no CPU path");
    if (!states || !locus_offsets || !out_states || !out_offsets || !out_weight || !out_npatterns || nloci < 1 || ntaxa < 1 ||
        ncols_total < 0)
        return fail(TPHIP_ERR_INVALID, "bad arguments");
    if (ncols_total >= ((int64_t)1 << 31)) return fail(TPHIP_ERR_INVALID, "too many columns for one call (2^31)");
    if (nloci >= ((int64_t)1 << (64 - kPatHashBits))) return fail(TPHIP_ERR_INVALID, "too many loci for one call");
    if (locus_offsets[0] != 0 || locus_offsets[nloci] != ncols_total) return fail(TPHIP_ERR_INVALID, "locus_offsets must span the columns");
    for (int64_t l = 0; l < nloci; ++l)
        if (locus_offsets[l + 1] < locus_offsets[l]) return fail(TPHIP_ERR_INVALID, "locus_offsets must not decrease");
    if (ncols_total == 0) {
        for (int64_t l = 0; l <= nloci; ++l) out_offsets[l] = 0;
        *out_npatterns = 0;
        return TPHIP_OK;
    }
    HIP_TRY(hipSetDevice(device));
    Scratch S, K;
    const size_t n = (size_t)ncols_total;
    uint8_t* d_s = S.get<uint8_t>(n * (size_t)ntaxa);
    int64_t* d_off = S.get<int64_t>((size_t)nloci + 1);
    if (!d_s || !d_off) return fail(TPHIP_ERR_HIP, "hipMalloc failed");
    HIP_TRY(hipMemcpy(d_s, states, n * (size_t)ntaxa, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(d_off, locus_offsets, sizeof(int64_t) * ((size_t)nloci + 1), hipMemcpyHostToDevice));
    uint8_t* d_out = nullptr; int64_t* d_newoff = nullptr; double* d_w = nullptr; int64_t* d_map = nullptr; int64_t npat = 0;
    int rc = compress_core(d_s, ncols_total, ntaxa, d_off, nloci, out_map != nullptr, S, K, &d_out, &d_newoff, &d_w, &d_map, &npat);
    if (rc) return rc;
    HIP_TRY(hipMemcpy(out_states, d_out, (size_t)npat * (size_t)ntaxa, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(out_offsets, d_newoff, sizeof(int64_t) * ((size_t)nloci + 1), hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(out_weight, d_w, sizeof(double) * (size_t)npat, hipMemcpyDeviceToHost));
    if (out_map) HIP_TRY(hipMemcpy(out_map, d_map, sizeof(int64_t) * n, hipMemcpyDeviceToHost));
    *out_npatterns = npat;
    return TPHIP_OK;
}

int tphip_state_histogram(int32_t device, const uint8_t* states, int64_t ncols_total, int32_t ntaxa,
                          const int64_t* locus_offsets, int64_t nloci, int64_t* hist) {
    if (tphip_device_count() <= 0) return fail(TPHIP_ERR_NO_DEVICE, "no HIP device visible: libtphip has no CPU path");
    if (!states || !locus_offsets || !hist || nloci < 1 || ntaxa < 1 || ncols_total < 0) return fail(TPHIP_ERR_INVALID, "bad arguments");
    HIP_TRY(hipSetDevice(device));
    Scratch S;
    const size_t nb = (size_t)ncols_total * (size_t)ntaxa;
    uint8_t* d_s = S.get<uint8_t>(nb);
    int64_t* d_o = S.get<int64_t>((size_t)nloci + 1);
    int64_t* d_h = S.get<int64_t>(16 * (size_t)nloci);
    if (!d_s || !d_o || !d_h) return fail(TPHIP_ERR_HIP, "hipMalloc failed");
    HIP_TRY(hipMemcpy(d_s, states, nb, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(d_o, locus_offsets, sizeof(int64_t) * ((size_t)nloci + 1), hipMemcpyHostToDevice));
    int rc = tphip_state_histogram_dev(device, d_s, ncols_total, ntaxa, d_o, nloci, d_h, nullptr);
    if (rc) return rc;
    HIP_TRY(hipMemcpy(hist, d_h, sizeof(int64_t) * 16 * (size_t)nloci, hipMemcpyDeviceToHost));
    return TPHIP_OK;
}

}  // extern "C"
