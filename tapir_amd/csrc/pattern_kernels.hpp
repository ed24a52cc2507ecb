// pattern_kernels.hpp -- collapse identical alignment columns of a locus into unique site patterns with counts.
//
// HyPhy never evaluates a column twice: `GetDataInfo(dupInfo, filteredData)` (tapir/data/models_and_rates.bf:960-963)
// maps every column to its unique pattern, stage 1 sums pattern log-likelihoods weighted by their counts and
// stage 2 fits one rate per pattern and reports it for every column that carries it (bf:1033-1070).  Same idea
// here, for the whole batch of loci at once:
//
//   pack_hash_kernel : one thread per column walks the taxa (coalesced byte rows), packs the normalised 4-bit state
//                      masks eight to a word into a column-major row [col][nwords] and hashes the row twice;
//                      sort key = (locus << 40) | 40 bits of hash 1, so that one radix sort groups equal columns
//                      of the same locus next to each other while keeping the loci in order;
//   (rocPRIM radix sort of (key, column) pairs -- stable, so equal columns stay in column order)
//   head_flag_kernel : a sorted position starts a new pattern unless key, second hash AND the full packed row equal
//                      its predecessor's (exact: a hash collision can only cost compression, never merge columns);
//   (rocPRIM inclusive scan of the flags = pattern index + 1)
//   scatter_kernel   : pattern p's states (unpacked back to taxon-major bytes), its count, and the column -> pattern map.
//
// HBM-bound byte/integer work: ntaxa bytes read + ntaxa/2 bytes written per column by the pack pass, then
// O(ntaxa/2) bytes per column for the comparisons; the sort moves 12 bytes per column per radix pass.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

namespace tphip {

constexpr int kPatHashBits = 40;

__device__ inline uint64_t pat_mix(uint64_t h, uint64_t w, uint64_t mul) {
    h ^= w;
    h *= mul;
    h ^= h >> 29;
    return h;
}

// locus of column c by binary search in offsets[0..nloci]
__device__ inline int64_t pat_locus_of(const int64_t* off, int64_t nloci, int64_t c) {
    int64_t lo = 0, hi = nloci;   // invariant: off[lo] <= c < off[hi]
    while (hi - lo > 1) {
        const int64_t mid = (lo + hi) >> 1;
        if (off[mid] <= c) lo = mid; else hi = mid;
    }
    return lo;
}

__global__ __launch_bounds__(256) void pack_hash_kernel(const uint8_t* states, int64_t ncols, int32_t ntaxa, int32_t nwords,
                                                       const int64_t* off, int64_t nloci, uint32_t* packed,
                                                       uint64_t* key, uint64_t* hash2, uint32_t* col_index) {
    const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= ncols) return;
    uint64_t h1 = 0x243F6A8885A308D3ull, h2 = 0x13198A2E03707344ull;
    for (int w = 0; w < nwords; ++w) {
        uint32_t word = 0;
        const int t0 = w * 8;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int t = t0 + j;
            unsigned m = 0;
            if (t < ntaxa) {
                m = states[(int64_t)t * ncols + c] & 15u;
                m = m ? m : 15u;
            }
            word |= m << (4 * j);
        }
        packed[c * nwords + w] = word;
        h1 = pat_mix(h1, word, 0x9E3779B97F4A7C15ull);
        h2 = pat_mix(h2, word, 0xC2B2AE3D27D4EB4Full);
    }
    h1 ^= h1 >> 32;
    h2 ^= h2 >> 31;
    const uint64_t locus = (uint64_t)pat_locus_of(off, nloci, c);
    key[c] = (locus << kPatHashBits) | (h1 & ((1ull << kPatHashBits) - 1));
    hash2[c] = h2;
    col_index[c] = (uint32_t)c;
}

__global__ __launch_bounds__(256) void head_flag_kernel(const uint64_t* key_sorted, const uint32_t* col_sorted,
                                                       const uint64_t* hash2, const uint32_t* packed, int32_t nwords,
                                                       int64_t ncols, int32_t* head) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= ncols) return;
    int h = 1;
    if (i > 0 && key_sorted[i] == key_sorted[i - 1]) {
        const uint32_t a = col_sorted[i], b = col_sorted[i - 1];
        if (hash2[a] == hash2[b]) {
            bool same = true;
            for (int w = 0; w < nwords; ++w) same &= (packed[(int64_t)a * nwords + w] == packed[(int64_t)b * nwords + w]);
            h = same ? 0 : 1;
        }
    }
    head[i] = h;
}

__global__ __launch_bounds__(256) void pattern_scatter_kernel(const uint32_t* col_sorted, const int32_t* head, const int64_t* incl,
                                                             const uint32_t* packed, int32_t nwords, int32_t ntaxa,
                                                             int64_t ncols, int64_t npat, uint8_t* out_states,
                                                             int32_t* count, int64_t* map) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= ncols) return;
    const int64_t p = incl[i] - 1;
    const uint32_t c = col_sorted[i];
    if (map) map[c] = p;
    atomicAdd(&count[p], 1);
    if (head[i]) {
        for (int w = 0; w < nwords; ++w) {
            const uint32_t word = packed[(int64_t)c * nwords + w];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int t = w * 8 + j;
                if (t < ntaxa) out_states[(int64_t)t * npat + p] = (uint8_t)((word >> (4 * j)) & 15u);
            }
        }
    }
}

// new_off[l] = pattern index of the first sorted position of locus l (the loci keep their order and sizes through the sort)
__global__ void pattern_offsets_kernel(const int64_t* off, int64_t nloci, int64_t ncols, const int64_t* incl, int64_t npat,
                                       int64_t* new_off) {
    const int64_t l = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (l > nloci) return;
    const int64_t o = off[l];
    new_off[l] = (o < ncols) ? incl[o] - 1 : npat;
}

__global__ void count_to_weight_kernel(const int32_t* count, int64_t npat, double* weight) {
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p < npat) weight[p] = (double)count[p];
}

}  // namespace tphip
