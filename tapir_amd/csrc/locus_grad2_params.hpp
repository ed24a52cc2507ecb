// locus_grad2_params.hpp -- parameter block, program records and launchers of the transition-matrix gradient kernel
// (locus_grad2_kernel.hpp); the kernels themselves are compiled in locus_grad2_launch.hip.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include <string>
#include <vector>
#include "gtr_model.hpp"
#include "locus_lik_common.hpp"
#include "locus_value_params.hpp"

namespace tphip {

constexpr int kGrad2Block = 128;
constexpr int kGrad2Waves = kGrad2Block / 64;
constexpr int kGrad2EF = 12;        // per (candidate, node): e^{lam_k t}[4], F01 F02 F03 F12 F13 F23, t, pad
constexpr int kGrad2MaxRDepth = 8;  // parked adjoints of the reverse sweep (LDS): log2(taxa) on a balanced tree

// reverse-sweep record of one internal node n with children A and B (binary trees), two int4:
//   a.x  flags (below)          a.y  node n (its tables; unused for the root)
//   a.z  child A: tape slot of its message (internal) or byte offset of its taxon's rows in the LDS tip table (tip)
//   a.w  child B: same
//   b.x  child A a tip: word index | bit position << 16 of its state code in the packed words
//   b.y  child B: same          b.z  node of child A | node of child B << 16
// Nodes come in pre-order.  The adjoint of an internal child A is carried in registers to the next record (A's own); when
// both children are internal B's adjoint is parked on a stack in LDS and popped by B's record, which follows A's subtree.
enum : int32_t { G2_IS_ROOT = 1, G2_A_TIP = 2, G2_B_TIP = 4, G2_POP_U = 8, G2_PUSH_B = 16 };

struct Grad2Params {
    const uint32_t* packed;        // [nwords][ncols_total] state codes, 8 per word in tip order (value_pack_codes_kernel)
    int64_t ncols_total;
    const int64_t* locus_offsets;  // [nloci+1]
    const double* col_weight;      // multiplicity of each column, null = 1
    const LocusModel* models;      // pi of the candidate's locus
    const int4* fops;              // forward op stream = locus_value_kernel's, BRANCH records carry their tape slot in .z
    const int4* rops;              // reverse records, 2 per internal node, pre-order
    int32_t nrops;                 // internal nodes
    int32_t ntaxa, nnodes, nwords;
    int32_t ntape;                 // tape slots = internal nodes below the root
    int32_t rdepth;                // parked adjoints
    const int32_t* tip_node;       // [ntaxa]
    const int32_t* cand_locus;     // [ncand]
    const double* eig;             // [ncand][36] lam, U, U^-1 (lik_eigen_kernel)
    const double* pmat;            // [ncand][nnodes][16] transposed transition matrices (lik_pmat_kernel)
    const double* ef;              // [ncand][nnodes][kGrad2EF]
    int64_t ncand;
    int32_t nsplit;
    double* tape;                  // [gridDim.x][ntape][4][kGrad2Block] messages of the internal branches
    // outputs per work item = cand * nsplit + slice (partials when nsplit > 1), as locus_grad_kernel's
    double* out_lnl;
    double* out_dexch;             // [items][6]
    double* out_dlogt;             // [items][nnodes] or null
    double* out_sum_dlogt;         // [items]
    double* out_d2logt;            // [items][nnodes] or null
};

// reverse program of a binary tree (host).  `tape_slot[node]` = slot of the node's message (-1: tip or root);
// `tip_pos[taxon]` = index of the taxon's tip in op order (its state code sits in word pos / 8 at bit 4 * (pos % 8)).
// Returns "" on success.
inline std::string build_grad2_program(int32_t nnodes, const int32_t* parent, const int32_t* leaf_taxon, const std::vector<int32_t>& tape_slot,
                                       const std::vector<int32_t>& tip_pos, std::vector<int4>* rops, int32_t* rdepth) {
    std::vector<std::vector<int32_t>> kids(nnodes);
    for (int32_t n = 0; n < nnodes; ++n) if (parent[n] >= 0) kids[parent[n]].push_back(n);
    for (int32_t n = 0; n < nnodes; ++n)
        if (!kids[n].empty() && kids[n].size() != 2) return "not a binary tree";
    std::vector<int32_t> need(nnodes, 0);   // parked adjoints needed below (and at) the node
    for (int32_t n = 0; n < nnodes; ++n) {  // post-order: children first
        if (kids[n].empty()) continue;
        int32_t a = kids[n][0], b = kids[n][1];
        const bool ta = kids[a].empty(), tb = kids[b].empty();
        if (ta && !tb) std::swap(a, b);                                  // a single internal child is A
        else if (!ta && !tb && need[a] > need[b]) std::swap(a, b);       // both internal: the shallower one first
        kids[n][0] = a; kids[n][1] = b;
        if (!kids[a].empty() && !kids[b].empty()) need[n] = std::max(need[a] + 1, need[b]);
        else if (!kids[a].empty()) need[n] = need[a];
    }
    *rdepth = need[nnodes - 1];
    rops->clear();
    struct Item { int32_t node; bool pop; };
    std::vector<Item> st;
    st.push_back({nnodes - 1, false});
    while (!st.empty()) {
        const Item it = st.back();
        st.pop_back();
        const int32_t n = it.node, a = kids[n][0], b = kids[n][1];
        const bool ta = kids[a].empty(), tb = kids[b].empty();
        int32_t flags = (parent[n] < 0 ? G2_IS_ROOT : 0) | (ta ? G2_A_TIP : 0) | (tb ? G2_B_TIP : 0) | (it.pop ? G2_POP_U : 0);
        if (!ta && !tb) flags |= G2_PUSH_B;
        auto where = [&](int32_t c, bool tip) { return tip ? leaf_taxon[c] * kValueTipRow * 8 : tape_slot[c]; };
        auto code_at = [&](int32_t c, bool tip) { return tip ? ((tip_pos[leaf_taxon[c]] >> 3) | ((4 * (tip_pos[leaf_taxon[c]] & 7)) << 16)) : 0; };
        rops->push_back(make_int4(flags, n, where(a, ta), where(b, tb)));
        rops->push_back(make_int4(code_at(a, ta), code_at(b, tb), a | (b << 16), 0));
        if (!tb) st.push_back({b, !ta});   // B after A's whole subtree: popped from the parked adjoints when both are internal
        if (!ta) st.push_back({a, false});
    }
    return "";
}

// host side (locus_grad2_launch.hip)
hipError_t launch_locus_grad2_kernel(int depth, dim3 grid, size_t lds_bytes, hipStream_t st, const Grad2Params* d_params);
hipError_t locus_grad2_kernel_allow_lds(int depth, size_t lds_bytes);
hipError_t locus_grad2_kernel_occupancy(int depth, size_t lds_bytes, int* blocks_per_cu);
hipError_t launch_grad2_ef_kernel(hipStream_t st, const double* eig, const double* blen_vecs, const int32_t* cand_vec, const double* cand_scale,
                                  const int32_t* cand_pidx, const double* cand_pfac, int64_t ncand, int32_t nnodes, double* ef);

}  // namespace tphip
