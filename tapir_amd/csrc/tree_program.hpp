// tree_program.hpp -- compile a rooted tree into a linear "pruning program" for the site-rate kernel.
//
// HyPhy evaluates LikelihoodFunction(siteFilter, siteTree) by a post-order (Felsenstein pruning)
// traversal (tapir/data/models_and_rates.bf:1003-1013, 1053).  On the GPU every lane owns one alignment
// column and all lanes walk the SAME tree, so the traversal is compiled once on the host into a short
// uniform instruction stream that the kernel reads through scalar loads:
//
//   TIP_SET  taxon,t   acc  = P(t*s) * tip(taxon)                 (start a new subtree in the accumulator)
//   TIP_MUL  taxon,t   acc *= P(t*s) * tip(taxon)                 (sibling tip folded into the accumulator)
//   BRANCH   t         acc  = P(t*s) * acc                        (finish an internal node, climb its branch)
//   PUSH               LDS stack <- acc                           (park a finished sibling)
//   POP_MUL            acc *= LDS stack                           (combine with the parked sibling)
//
// "acc" is the (value, d/du, d2/du2) triple of the 4-state partial likelihood vector, u = log(siteRate).
// Children are visited in Sethi-Ullman order (deepest internal child first, tips last) so that the number
// of partials parked in LDS at any time -- the stack depth -- is minimal: 0 for a caterpillar,
// log2(N)-1 for a perfectly balanced tree.
#pragma once
#include <algorithm>
#include <cstdint>
#include <string>
#include <vector>

namespace tphip {

enum : int32_t { OP_TIP_SET = 0, OP_TIP_MUL = 1, OP_BRANCH = 2, OP_PUSH = 3, OP_POP_MUL = 4,
                 OP_CHERRY = 5 };  // fused stream only: TIP_SET + TIP_MUL on equally long branches
// fused stream only, OR-ed into the code: the interpreter pays ~35 scalar instructions and several branches per op,
// so the two cheap stack ops ride on their neighbours
enum : int32_t { OP_CODE_MASK = 0xff,
                 OP_PUSH_BEFORE = 0x100,   // on TIP_SET / CHERRY: park the accumulator first (the PUSH that preceded it)
                 OP_POP_AFTER = 0x200 };   // on BRANCH: multiply the parked sibling in afterwards (the POP_MUL that followed)

struct TreeOp {
    int32_t code;
    int32_t taxon;  // alignment row for TIP_* ops
    double t;       // branch length (already / correction) for TIP_* and BRANCH
};
static_assert(sizeof(TreeOp) == 16, "TreeOp is read with one s_load_dwordx4");

struct TreeProgram {
    std::vector<TreeOp> ops;
    // The same program with every TIP_SET + TIP_MUL pair on equally long branches (the two tips of a cherry in a
    // chronogram) folded into one CHERRY op: both messages share exp(lambda_k t s).  Tip order is unchanged, so the
    // packed tip words of classify_kernel serve both streams.  Only site_rate_kernel's packed path reads it.
    std::vector<TreeOp> fused_ops;
    std::vector<int32_t> op_node;  // tree node whose branch a TIP_* / BRANCH op climbs (-1 for PUSH / POP_MUL)
    std::vector<int32_t> op_tape;     // reverse-mode tape slot written by a BRANCH / PUSH op (-1 otherwise)
    std::vector<int32_t> op_partner;  // POP_MUL: tape slot of the PUSH it pops (-1 otherwise)
    int32_t ntape = 0;
    int32_t stack_depth = 0;   // LDS slots per lane
    double chrono_length = 0;  // sum of all branch lengths (bf:1006-1013)
    int32_t nleaves = 0;
};

// Returns "" on success, else an error message.
inline std::string build_tree_program(int32_t ntaxa, int32_t nnodes, const int32_t* parent, const double* blen,
                                      const int32_t* leaf_taxon, TreeProgram* out) {
    if (nnodes < 3) return "tree needs at least two leaves";
    std::vector<std::vector<int32_t>> kids(nnodes);
    int32_t root = -1;
    for (int32_t n = 0; n < nnodes; ++n) {
        int32_t p = parent[n];
        if (p < 0) {
            if (root >= 0) return "tree has more than one root";
            root = n;
        } else {
            if (p >= nnodes || p <= n) return "tree nodes must be in post-order (parent index > child index)";
            kids[p].push_back(n);
        }
    }
    if (root != nnodes - 1) return "root must be the last node";
    std::vector<char> seen(ntaxa > 0 ? ntaxa : 0, 0);
    out->chrono_length = 0;
    out->nleaves = 0;
    for (int32_t n = 0; n < nnodes; ++n) {
        bool leaf = kids[n].empty();
        if (leaf) {
            int32_t tx = leaf_taxon[n];
            if (tx < 0 || tx >= ntaxa) return "leaf_taxon out of range";
            if (seen[tx]) return "two leaves map to the same alignment row";
            seen[tx] = 1;
            ++out->nleaves;
        } else if (leaf_taxon[n] >= 0) {
            return "internal node carries a taxon";
        }
        if (n != root) {
            if (!(blen[n] >= 0.0)) return "negative or NaN branch length";
            out->chrono_length += blen[n];
        }
    }
    if (kids[root].empty()) return "root has no children";
    // Sethi-Ullman need: slots (accumulator included) to evaluate the subtree; a tip needs none of its own.
    std::vector<int32_t> need(nnodes, 0);
    for (int32_t n = 0; n < nnodes; ++n) {  // post-order: children first
        if (kids[n].empty()) continue;
        std::stable_sort(kids[n].begin(), kids[n].end(), [&](int32_t a, int32_t b) { return need[a] > need[b]; });
        int32_t nd = 1;
        for (size_t i = 0; i < kids[n].size(); ++i) {
            int32_t c = kids[n][i];
            int32_t extra = (i == 0 || need[c] == 0) ? 0 : 1;  // a later internal child runs beside a parked partial
            nd = std::max(nd, need[c] + extra);
        }
        need[n] = nd;
    }
    out->stack_depth = need[root] - 1;
    out->ops.clear();
    out->op_node.clear();
    // iterative emit (explicit stack: trees can be deep caterpillars)
    struct Frame { int32_t node; size_t next; };
    std::vector<Frame> st;
    st.push_back({root, 0});
    while (!st.empty()) {
        Frame& f = st.back();
        int32_t n = f.node;
        if (f.next < kids[n].size()) {
            int32_t c = kids[n][f.next];
            bool first = (f.next == 0);
            ++f.next;
            if (kids[c].empty()) {
                out->ops.push_back({first ? OP_TIP_SET : OP_TIP_MUL, leaf_taxon[c], blen[c]});
                out->op_node.push_back(c);
            } else {
                if (!first) { out->ops.push_back({OP_PUSH, 0, 0.0}); out->op_node.push_back(-1); }
                st.push_back({c, 0});
            }
        } else {
            st.pop_back();
            if (!st.empty()) {
                // finished internal child n of st.back(): climb its branch, then merge with a parked sibling
                out->ops.push_back({OP_BRANCH, 0, blen[n]});
                out->op_node.push_back(n);
                bool was_first = (st.back().next == 1);
                if (!was_first) { out->ops.push_back({OP_POP_MUL, 0, 0.0}); out->op_node.push_back(-1); }
            }
        }
    }
    // tape slots for the gradient kernel (locus_grad_kernel)
    out->op_tape.assign(out->ops.size(), -1);
    out->op_partner.assign(out->ops.size(), -1);
    out->ntape = 0;
    std::vector<int32_t> pushed;
    for (size_t i = 0; i < out->ops.size(); ++i) {
        const int32_t code = out->ops[i].code;
        if (code == OP_BRANCH) {
            out->op_tape[i] = out->ntape++;
        } else if (code == OP_PUSH) {
            out->op_tape[i] = out->ntape++;
            pushed.push_back(out->op_tape[i]);
        } else if (code == OP_POP_MUL) {
            if (pushed.empty()) return "internal error: POP_MUL without PUSH";
            out->op_partner[i] = pushed.back();
            pushed.pop_back();
        }
    }
    if (!pushed.empty()) return "internal error: unbalanced PUSH";
    out->fused_ops.clear();
    int32_t pending = 0;
    for (size_t i = 0; i < out->ops.size(); ++i) {
        const TreeOp& a = out->ops[i];
        const bool has_next = i + 1 < out->ops.size();
        if (a.code == OP_PUSH && has_next && out->ops[i + 1].code == OP_TIP_SET) {
            pending = OP_PUSH_BEFORE;   // carried by the TIP_SET / CHERRY that follows
        } else if (a.code == OP_TIP_SET && has_next && out->ops[i + 1].code == OP_TIP_MUL && out->ops[i + 1].t == a.t) {
            out->fused_ops.push_back({OP_CHERRY | pending, a.taxon, a.t});
            pending = 0;
            ++i;
        } else if (a.code == OP_BRANCH && has_next && out->ops[i + 1].code == OP_POP_MUL) {
            out->fused_ops.push_back({OP_BRANCH | OP_POP_AFTER, a.taxon, a.t});
            ++i;
        } else {
            out->fused_ops.push_back({a.code | pending, a.taxon, a.t});
            pending = 0;
        }
    }
    return "";
}

}  // namespace tphip
