// locus_value_params.hpp -- parameter block and launchers of the transition-matrix value kernels (locus_value_kernel.hpp);
// the kernels themselves are compiled in locus_value_launch.hip.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include "gtr_model.hpp"

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
