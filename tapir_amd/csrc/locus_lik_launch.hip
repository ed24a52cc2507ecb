// locus_lik_launch.hip -- translation unit of the stage-1 eigenbasis value kernel (fallback for deep trees) and of the
// reverse-mode gradient kernel.
#include <hip/hip_runtime.h>

#include "locus_lik_kernel.hpp"

namespace tphip {

hipError_t launch_locus_loglik_kernel(dim3 grid, size_t lds_bytes, hipStream_t st, const LikParams& L) {
    locus_loglik_kernel<<<grid, dim3(kLikBlock), lds_bytes, st>>>(L);
    return hipGetLastError();
}

hipError_t launch_locus_grad_kernel(dim3 grid, size_t lds_bytes, hipStream_t st, const GradParams* d_params) {
    locus_grad_kernel<<<grid, dim3(kGradBlock), lds_bytes, st>>>(d_params);
    return hipGetLastError();
}

hipError_t launch_split_sum_kernel(hipStream_t st, const double* part, double* out, int64_t ncand, int nsplit, int width) {
    const int64_t n = ncand * width;
    split_sum_kernel<<<dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st>>>(part, out, ncand, nsplit, width);
    return hipGetLastError();
}

hipError_t locus_loglik_kernel_allow_lds(size_t lds_bytes) {
    return hipFuncSetAttribute((const void*)locus_loglik_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
}

hipError_t locus_grad_kernel_allow_lds(size_t lds_bytes) {
    return hipFuncSetAttribute((const void*)locus_grad_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
}

hipError_t locus_grad_kernel_occupancy(size_t lds_bytes, int* blocks_per_cu) {
    return hipOccupancyMaxActiveBlocksPerMultiprocessor(blocks_per_cu, locus_grad_kernel, kGradBlock, lds_bytes);
}

}  // namespace tphip
