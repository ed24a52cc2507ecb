// locus_grad2_launch.hip -- translation unit of the transition-matrix gradient kernel (locus_grad2_kernel.hpp).
#include <hip/hip_runtime.h>

#include "locus_grad2_kernel.hpp"

namespace tphip {

template <int D> static const void* grad2_fn() { return (const void*)locus_grad2_kernel<D>; }

// depth of the forward sweep's register stack -> instantiation
#define TPHIP_GRAD2_DISPATCH(CALL)                                       \
    if (depth <= 2) { CALL(2); } else if (depth == 3) { CALL(3); }       \
    else if (depth == 4) { CALL(4); } else { CALL(5); }

hipError_t launch_locus_grad2_kernel(int depth, dim3 grid, size_t lds_bytes, hipStream_t st, const Grad2Params* d_params) {
    if (depth > kValueMaxDepth) return hipErrorInvalidValue;
#define TPHIP_GRAD2_LAUNCH(D) locus_grad2_kernel<D><<<grid, dim3(kGrad2Block), lds_bytes, st>>>(d_params)
    TPHIP_GRAD2_DISPATCH(TPHIP_GRAD2_LAUNCH)
#undef TPHIP_GRAD2_LAUNCH
    return hipGetLastError();
}

hipError_t locus_grad2_kernel_allow_lds(int depth, size_t lds_bytes) {
    if (depth > kValueMaxDepth) return hipErrorInvalidValue;
    const void* fn = nullptr;
#define TPHIP_GRAD2_FN(D) fn = grad2_fn<D>()
    TPHIP_GRAD2_DISPATCH(TPHIP_GRAD2_FN)
#undef TPHIP_GRAD2_FN
    return hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
}

hipError_t locus_grad2_kernel_occupancy(int depth, size_t lds_bytes, int* blocks_per_cu) {
    if (depth > kValueMaxDepth) return hipErrorInvalidValue;
    const void* fn = nullptr;
#define TPHIP_GRAD2_FN(D) fn = grad2_fn<D>()
    TPHIP_GRAD2_DISPATCH(TPHIP_GRAD2_FN)
#undef TPHIP_GRAD2_FN
    return hipOccupancyMaxActiveBlocksPerMultiprocessor(blocks_per_cu, fn, kGrad2Block, lds_bytes);
}

hipError_t launch_grad2_ef_kernel(hipStream_t st, const double* eig, const double* blen_vecs, const int32_t* cand_vec, const double* cand_scale,
                                  const int32_t* cand_pidx, const double* cand_pfac, int64_t ncand, int32_t nnodes, double* ef) {
    const int64_t n = ncand * nnodes;
    grad2_ef_kernel<<<dim3((unsigned)((n + 127) / 128)), dim3(128), 0, st>>>(eig, blen_vecs, cand_vec, cand_scale, cand_pidx, cand_pfac, ncand,
                                                                          nnodes, ef);
    return hipGetLastError();
}

}  // namespace tphip
