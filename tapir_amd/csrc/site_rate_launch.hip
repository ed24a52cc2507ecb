// site_rate_launch.hip -- the translation unit of the site-rate kernels.
//
// This file, and only this file, is compiled with `-mllvm -structurizecfg-skip-uniform-regions` (see
// __graft_entry__.build): the op interpreter of site_rate_kernel branches on wave-uniform op codes, and structurizing
// those branches adds Flow blocks whose phis keep the 12-double accumulator alive on every path (8-36 v_mov_b64 per op,
// DESIGN section 8 r1 v7).  The option is off by default in LLVM's AMDGPU pipeline for a reason: in round 2 it
// miscompiled classify_kernel as soon as that kernel got a wave-uniform branch inside a divergent if / else (two stores
// tail-merged across the paths with the address register of one path holding the other path's temporary: found with
// rocgdb, precise-memory mode).  So nothing else is built with it; whatever changes in site_rate_kernel.hpp is covered
// by the GPU-vs-oracle parity tests, which exercise every op and scheduling mode of these kernels.
#include <hip/hip_runtime.h>

#include "site_rate_kernel.hpp"

namespace tphip {

hipError_t launch_site_rate_kernel(int variant, dim3 grid, size_t lds_bytes, hipStream_t st, const SiteParams& S) {
    const dim3 block(kSiteBlock);
    if (variant == 0) site_rate_kernel<0><<<grid, block, lds_bytes, st>>>(S);
    else if (variant == 2) site_rate_kernel<2><<<grid, block, lds_bytes, st>>>(S);
    else if (variant == 8) site_rate_kernel<8><<<grid, block, lds_bytes, st>>>(S);
    else if (variant == kMixedVariant + 2) site_rate_kernel<2, false, true><<<grid, block, lds_bytes, st>>>(S);
    else if (variant == kMixedVariant + 8) site_rate_kernel<8, false, true><<<grid, block, lds_bytes, st>>>(S);
    else if (variant == kStreamWords) site_rate_kernel<kStreamWords><<<grid, block, lds_bytes, st>>>(S);
    else if (variant == kStreamWordsSpill) site_rate_kernel<kStreamWords, true><<<grid, block, lds_bytes, st>>>(S);
    else return hipErrorInvalidValue;
    return hipGetLastError();
}

hipError_t site_rate_kernel_occupancy(int variant, size_t lds_bytes, int* blocks_per_cu) {
    if (variant == 0) return hipOccupancyMaxActiveBlocksPerMultiprocessor(blocks_per_cu, site_rate_kernel<0>, kSiteBlock, lds_bytes);
    if (variant == 2) return hipOccupancyMaxActiveBlocksPerMultiprocessor(blocks_per_cu, site_rate_kernel<2>, kSiteBlock, lds_bytes);
    if (variant == 8) return hipOccupancyMaxActiveBlocksPerMultiprocessor(blocks_per_cu, site_rate_kernel<8>, kSiteBlock, lds_bytes);
    if (variant == kMixedVariant + 2) return hipOccupancyMaxActiveBlocksPerMultiprocessor(blocks_per_cu, site_rate_kernel<2, false, true>, kSiteBlock, lds_bytes);
    if (variant == kMixedVariant + 8) return hipOccupancyMaxActiveBlocksPerMultiprocessor(blocks_per_cu, site_rate_kernel<8, false, true>, kSiteBlock, lds_bytes);
    if (variant == kStreamWords)
        return hipOccupancyMaxActiveBlocksPerMultiprocessor(blocks_per_cu, site_rate_kernel<kStreamWords>, kSiteBlock, lds_bytes);
    if (variant == kStreamWordsSpill)
        return hipOccupancyMaxActiveBlocksPerMultiprocessor(blocks_per_cu, site_rate_kernel<kStreamWords, true>, kSiteBlock, lds_bytes);
    return hipErrorInvalidValue;
}

hipError_t launch_eval_columns_kernel(dim3 grid, size_t lds_bytes, hipStream_t st, const EvalParams& E) {
    eval_columns_kernel<<<grid, dim3(kSiteBlock), lds_bytes, st>>>(E);
    return hipGetLastError();
}

hipError_t launch_scan_counts_kernel(hipStream_t st, const int32_t* count, int64_t nloci, int32_t chunk_cols, int64_t* prefix,
                                     int64_t* slice_prefix) {
    scan_counts_kernel<<<dim3(1), dim3(1024), 0, st>>>(count, nloci, chunk_cols, prefix, slice_prefix);
    return hipGetLastError();
}

}  // namespace tphip
