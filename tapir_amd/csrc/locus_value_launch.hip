// locus_value_launch.hip -- translation unit of the stage-1 value kernels on transition matrices.
#include <hip/hip_runtime.h>

#include "locus_value_kernel.hpp"

namespace tphip {

template <int C, int D> static const void* value_fn() { return (const void*)locus_value_kernel<C, D>; }

// (cols, depth) -> instantiation with the smallest register stack that holds the tree's parked siblings
#define TPHIP_VALUE_DISPATCH(CALL)                                             \
    if (cols == 1) {                                                           \
        if (depth <= 2) { CALL(1, 2); } else if (depth == 3) { CALL(1, 3); }   \
        else if (depth == 4) { CALL(1, 4); } else { CALL(1, 5); }              \
    } else {                                                                   \
        if (depth <= 2) { CALL(2, 2); } else if (depth == 3) { CALL(2, 3); }   \
        else if (depth == 4) { CALL(2, 4); } else { CALL(2, 5); }              \
    }

hipError_t launch_locus_value_kernel(int cols, int depth, dim3 grid, size_t lds_bytes, hipStream_t st, const ValueParams& V) {
    if ((cols != 1 && cols != 2) || depth > kValueMaxDepth) return hipErrorInvalidValue;
#define TPHIP_VALUE_LAUNCH(C, D) locus_value_kernel<C, D><<<grid, dim3(kLikBlock), lds_bytes, st>>>(V)
    TPHIP_VALUE_DISPATCH(TPHIP_VALUE_LAUNCH)
#undef TPHIP_VALUE_LAUNCH
    return hipGetLastError();
}

hipError_t locus_value_kernel_allow_lds(int cols, int depth, size_t lds_bytes) {
    if ((cols != 1 && cols != 2) || depth > kValueMaxDepth) return hipErrorInvalidValue;
    const void* fn = nullptr;
#define TPHIP_VALUE_FN(C, D) fn = value_fn<C, D>()
    TPHIP_VALUE_DISPATCH(TPHIP_VALUE_FN)
#undef TPHIP_VALUE_FN
    return hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
}

hipError_t launch_value_pack_codes_kernel(hipStream_t st, const uint8_t* states, int64_t ncols_total, const int32_t* tip_taxon,
                                         int32_t nwords, uint32_t* packed) {
    if (ncols_total <= 0) return hipSuccess;
    value_pack_codes_kernel<<<dim3((unsigned)((ncols_total + 255) / 256)), dim3(256), 0, st>>>(states, ncols_total, tip_taxon, nwords, packed);
    return hipGetLastError();
}

hipError_t launch_lik_eigen_kernel(hipStream_t st, const LocusModel* models, const int32_t* cand_locus, const double* cand_exch,
                                   int64_t ncand, double* eig_out) {
    lik_eigen_kernel<<<dim3((unsigned)((ncand + 63) / 64)), dim3(64), 0, st>>>(models, cand_locus, cand_exch, ncand, eig_out);
    return hipGetLastError();
}

hipError_t launch_lik_pmat_kernel(hipStream_t st, const double* eig, const double* blen_vecs, const int32_t* cand_vec,
                                  const double* cand_scale, const int32_t* cand_pidx, const double* cand_pfac, int64_t ncand,
                                  int32_t nnodes, double* pmat) {
    const int64_t n = ncand * nnodes;
    lik_pmat_kernel<<<dim3((unsigned)((n + 127) / 128)), dim3(128), 0, st>>>(eig, blen_vecs, cand_vec, cand_scale, cand_pidx, cand_pfac,
                                                                          ncand, nnodes, pmat);
    return hipGetLastError();
}

}  // namespace tphip
