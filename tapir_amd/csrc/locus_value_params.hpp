// locus_value_params.hpp -- parameter block and launchers of the transition-matrix value kernels (locus_value_kernel.hpp);
// the kernels themselves are compiled in locus_value_launch.hip.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include <vector>
#include "gtr_model.hpp"
#include "tree_program.hpp"

namespace tphip {

constexpr size_t kValueWorkspaceBytes = (size_t)1 << 30;
constexpr int kValueOpEnd = 0xff;   // code of the two records that close the op stream
constexpr int kValueTipRow = 20;     // doubles per taxon in the LDS tip table: 4 rows of P^T + a row of ones
// Tip states travel as 4-bit CODES, not masks: 0..3 = A, C, G, T (row of the tip table), 4 = gap / N (the row of ones),
// 5..14 = the ten other IUPAC sets (slow path: the rows of the set's bits are added).  16 nibbles each:
constexpr unsigned long long kValueCodeOfMask = 0x4EDCBA9387625104ull;   // nibble m = code of state mask m (0 and 15 -> 4)
constexpr unsigned long long kValueMaskOfCode = 0x0EDCBA97653F8421ull;   // nibble c = state mask of code c

struct ValueParams {
    const uint8_t* states;         // [ntaxa][ncols_total]
    int64_t ncols_total;
    const int64_t* locus_offsets;  // [nloci+1]
    const double* col_weight;      // multiplicity of each column, null = 1
    const LocusModel* models;      // pi of the candidate's locus (root frequencies)
    // fused op stream, closed by two kValueOpEnd records; everything an op needs is precomputed in its record:
    //   x = code | flags | shA << 12 | fetchA << 17 | shB << 20 | fetchB << 25   (sh = bit position of the tip's state code in
    //       its packed word, fetch = the tip opens a new word)
    //   y = tips: byte offset of taxon A's rows in the LDS tip table;  BRANCH: byte offset of the node's matrix
    //   z = CHERRY: byte offset of taxon B's rows
    //   w = word index of tip A | word index of tip B << 16
    const int4* vops;
    const int32_t* tip_taxon;      // taxon of the j-th tip, in op order, padded to a multiple of 8
    const int32_t* tip_node;       // [ntaxa] tree node of the taxon's tip (-1: the taxon is not in the tree)
    const uint32_t* packed;        // [nwords][ncols_total] the same state codes, 8 per word in tip order, packed once when the
                                   // library uploaded the alignment itself (value_pack_codes_kernel); null: the kernel packs
                                   // its columns from `states` at the start of every column group
    int32_t nvops, ntaxa, nnodes, nwords;
    const int32_t* cand_locus;     // [ncand of this launch]
    const double* pmat;            // [ncand of this launch][nnodes][16]
    double* out;                   // [ncand * nsplit]
    int32_t nsplit;
};

// The value kernels' fused op stream from the plain tree program (host): TIP_SET + TIP_MUL -> CHERRY (whatever the branch
// lengths: the messages are table rows), PUSH carried by the TIP_SET / CHERRY that follows it, POP_MUL by the BRANCH it
// follows; closed by two kValueOpEnd records.  tip_node[taxon] = tree node of the taxon's tip.  False: the program has a
// shape the stream cannot express (the eigenbasis kernels run instead).
inline bool build_value_program(const TreeProgram& prog, int32_t ntaxa, int32_t nnodes, std::vector<int4>* vops_out,
                                std::vector<int32_t>* tip_node_out) {
    const auto& ops = prog.ops;
    const auto& node = prog.op_node;
    std::vector<int4>& vops = *vops_out;
    std::vector<int32_t>& tip_node = *tip_node_out;
    vops.clear();
    tip_node.assign(ntaxa, -1);
    int32_t pending = 0, tip = 0;
    bool ok = nnodes < 65536 && (int64_t)ntaxa * kValueTipRow * 8 < (1 << 30);
    auto tip_bits = [&](int shift_pos, int fetch_pos) { return ((4 * (tip & 7)) << shift_pos) | (((tip & 7) == 0 ? 1 : 0) << fetch_pos); };
    for (size_t i = 0; i < ops.size() && ok; ++i) {
        const int32_t code = ops[i].code;
        const bool has_next = i + 1 < ops.size();
        if (code == OP_PUSH) {
            if (!(has_next && ops[i + 1].code == OP_TIP_SET) || pending) ok = false;   // (a PUSH is always followed by a TIP_SET)
            pending = OP_PUSH_BEFORE;
        } else if (code == OP_TIP_SET && has_next && ops[i + 1].code == OP_TIP_MUL) {
            int32_t x = OP_CHERRY | pending | tip_bits(12, 17);
            const int32_t wa = tip >> 3;
            ++tip;
            x |= tip_bits(20, 25);
            const int32_t wb = tip >> 3;
            ++tip;
            vops.push_back(make_int4(x, ops[i].taxon * kValueTipRow * 8, ops[i + 1].taxon * kValueTipRow * 8, wa | (wb << 16)));
            tip_node[ops[i].taxon] = node[i];
            tip_node[ops[i + 1].taxon] = node[i + 1];
            pending = 0;
            ++i;
        } else if (code == OP_TIP_SET || code == OP_TIP_MUL) {
            vops.push_back(make_int4(code | pending | tip_bits(12, 17), ops[i].taxon * kValueTipRow * 8, 0, tip >> 3));
            tip_node[ops[i].taxon] = node[i];
            ++tip;
            pending = 0;
        } else if (code == OP_BRANCH) {
            const bool pop = has_next && ops[i + 1].code == OP_POP_MUL;
            vops.push_back(make_int4(OP_BRANCH | (pop ? OP_POP_AFTER : 0), node[i] * 128, 0, 0));
            if (pop) ++i;
        } else {
            ok = false;   // a POP_MUL that does not follow a BRANCH
        }
    }
    vops.push_back(make_int4(kValueOpEnd, 0, 0, 0));
    vops.push_back(make_int4(kValueOpEnd, 0, 0, 0));
    return ok;
}

constexpr int kValueMaxDepth = 5;   // deepest register stack instantiated (a Sethi-Ullman-ordered tree of 2^(D+1) tips needs D)

// host side of the value kernels (locus_value_launch.hip)
hipError_t launch_locus_value_kernel(int cols, int depth, dim3 grid, size_t lds_bytes, hipStream_t st, const ValueParams& V);
hipError_t locus_value_kernel_allow_lds(int cols, int depth, size_t lds_bytes);
hipError_t launch_value_pack_codes_kernel(hipStream_t st, const uint8_t* states, int64_t ncols_total, const int32_t* tip_taxon,
                                         int32_t nwords, uint32_t* packed);
hipError_t launch_lik_eigen_kernel(hipStream_t st, const LocusModel* models, const int32_t* cand_locus, const double* cand_exch,
                                   int64_t ncand, double* eig_out);
hipError_t launch_lik_pmat_kernel(hipStream_t st, const double* eig, const double* blen_vecs, const int32_t* cand_vec,
                                  const double* cand_scale, const int32_t* cand_pidx, const double* cand_pfac, int64_t ncand,
                                  int32_t nnodes, double* pmat);

}  // namespace tphip
