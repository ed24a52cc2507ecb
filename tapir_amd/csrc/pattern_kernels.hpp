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
#include "tphip.h"

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

// ---------------------------------------------------------------------------------------------------------------------
// Stage 2 (per-site rates): optimise one rate per unique pattern (bf:1033-1044: `GetDataInfo(dupInfo...)`,
// `alreadyDone[siteMap]`) and copy it to the pattern's other columns.  Everything stays in the hot path's own pass:
//
//   classify_kernel       also leaves a 64-bit hash of every column's packed tip words (pi_kernels.hpp);
//   dedup_estimate_kernel one workgroup per locus: a 64 Ki-bit LDS bitmap of the hashes of the locus' first work columns
//                         estimates the number of distinct patterns (linear counting); the locus is de-duplicated only if that
//                         estimate is below kDedupWorthIt of its work columns (synthetic alignments with random gaps
//                         hardly repeat a column: they skip the rest of this machinery), and then clears its slice of
//                         the hash table;
//   dedup_insert_kernel   work column -> open-addressing table of its locus (atomicCAS on the hash, atomicMin on the
//                         column index: the representative of a pattern is its FIRST column, whatever the schedule);
//   dedup_resolve_kernel  a work column whose table entry names another column compares its packed words with that
//                         column's (exact: a hash collision only costs the saving) and, if equal, leaves the work
//                         list: flag = kFlagDuplicate, dup_of = representative;
//   (compact / scan / site_rate_kernel run on what is left)
//   dedup_scatter_kernel  copies (rate, subst, lnL, flag) from the representative.
//
// Results are bit-identical with and without de-duplication: a column's answer depends only on its packed words and
// its locus' model (tests: test_results_do_not_depend_on_how_the_work_list_is_shared_out, test_stage2_pattern_dedup).
// Byte/integer work, HBM- and atomics-bound: 8 B hash + 4 B dup_of per column, 24 B of table per column of a
// de-duplicated locus.
constexpr uint8_t kFlagDuplicate = 0x40;      // internal: never leaves launch_site_rates
constexpr unsigned long long kDedupEmpty = ~0ull;
constexpr int kDedupBitmapBits = 1 << 16;
constexpr double kDedupWorthIt = 0.85;
constexpr int kDedupMinColumns = 32;
enum : int32_t { DEDUP_AUTO = 0, DEDUP_OFF = 1, DEDUP_ON = 2 };

struct DedupParams {
    const int64_t* locus_offsets;   // [nloci+1]
    const int32_t* chunk_locus;     // 1024-column chunks (the PI chunk tables)
    const int32_t* chunk_index;
    const uint64_t* hash;           // [ncols] from classify_kernel
    const uint32_t* packed;         // [nwords][ncols]
    int32_t nwords;
    int64_t ncols_total;
    uint8_t* flag;                  // [ncols]
    int32_t* dup_of;                // [ncols] representative column, or -1
    unsigned long long* tab_key;    // [2 * ncols]; locus l owns [2 * lo, 2 * hi)
    int32_t* tab_val;               // [2 * ncols] smallest column (locus-relative) carrying the hash
    int32_t* on;                    // [nloci] 1 = this locus is de-duplicated in this launch
    int32_t mode;                   // DEDUP_*
    double* rate; double* subst; double* lnl;
};

__device__ __forceinline__ unsigned long long dedup_key(uint64_t h) { return h == kDedupEmpty ? 0ull : h; }

constexpr int64_t kDedupEstimatePrefix = 16384;   // columns of a locus the estimate looks at (0.17 -> 0.02 ms on C3)
// Batches below this many columns are not de-duplicated in automatic mode: they run in the latency-bound small-batch
// modes of site_rate_kernel (a launch lasts as long as its slowest wave), where fewer columns hardly shorten the kernel and the four extra
// launches cost more than they save (C2, 5e5 columns: 0.50 -> 0.53 ms per step with the machinery idling).
constexpr int64_t kDedupAutoMinColumns = 1 << 20;

// kBlock: 1024 threads for long loci, 256 for short ones (1000 workgroups of 1024 threads in front of the one-wave
// workgroups of site_rate_kernel made THAT kernel 24 % slower on C2 -- the effect round 1 saw with compact_kernel)
template <int kBlock>
__global__ __launch_bounds__(kBlock) void dedup_estimate_kernel(DedupParams P) {
    __shared__ unsigned bitmap[kDedupBitmapBits / 32];
    __shared__ int counts[2];
    const int locus = blockIdx.x;
    const int64_t lo = P.locus_offsets[locus], hi = P.locus_offsets[locus + 1];
    for (int i = threadIdx.x; i < kDedupBitmapBits / 32; i += blockDim.x) bitmap[i] = 0u;
    if (threadIdx.x < 2) counts[threadIdx.x] = 0;
    __syncthreads();
    // a prefix of the locus is enough for the estimate (repeats within 16 Ki columns; the bitmap's load stays below 0.25)
    const int64_t end = (hi - lo > kDedupEstimatePrefix) ? lo + kDedupEstimatePrefix : hi;
    int mine = 0;
    for (int64_t c = lo + threadIdx.x; c < end; c += blockDim.x) {
        if (P.flag[c] != TPHIP_FLAG_OK) continue;
        ++mine;
        const unsigned b = (unsigned)(P.hash[c] >> 17) & (kDedupBitmapBits - 1);
        atomicOr(&bitmap[b >> 5], 1u << (b & 31));
    }
    atomicAdd(&counts[0], mine);
    __syncthreads();
    int bits = 0;
    for (int i = threadIdx.x; i < kDedupBitmapBits / 32; i += blockDim.x) bits += __popc(bitmap[i]);
    atomicAdd(&counts[1], bits);
    __syncthreads();
    const int n = counts[0];
    const double fill = (double)counts[1] / (double)kDedupBitmapBits;
    const double distinct = (fill < 1.0) ? -(double)kDedupBitmapBits * log(1.0 - fill) : 1e300;   // linear counting
    const bool on = P.mode == DEDUP_ON || (P.mode == DEDUP_AUTO && n >= kDedupMinColumns && distinct <= kDedupWorthIt * (double)n);
    if (threadIdx.x == 0) P.on[locus] = on ? 1 : 0;
    if (on) {
        for (int64_t i = 2 * lo + threadIdx.x; i < 2 * hi; i += blockDim.x) { P.tab_key[i] = kDedupEmpty; P.tab_val[i] = 0x7fffffff; }
    }
}

// thread = column (1024-column chunks of one locus, 4 columns per thread strided by the block size)
__global__ __launch_bounds__(256) void dedup_insert_kernel(DedupParams P) {
    const int locus = P.chunk_locus[blockIdx.x];
    if (!P.on[locus]) return;
    const int64_t lo = P.locus_offsets[locus], hi = P.locus_offsets[locus + 1];
    const int64_t base = lo + (int64_t)P.chunk_index[blockIdx.x] * 1024;
    const unsigned long long size = 2ull * (unsigned long long)(hi - lo);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int64_t c = base + j * 256 + threadIdx.x;
        if (c >= hi || P.flag[c] != TPHIP_FLAG_OK) continue;
        const unsigned long long h = dedup_key(P.hash[c]);
        unsigned long long slot = h % size;
        for (;;) {   // the table is twice the locus: a free or matching slot always exists
            const unsigned long long prev = atomicCAS(&P.tab_key[2 * lo + slot], kDedupEmpty, h);
            if (prev == kDedupEmpty || prev == h) { atomicMin(&P.tab_val[2 * lo + slot], (int)(c - lo)); break; }
            slot = (slot + 1 == size) ? 0 : slot + 1;
        }
    }
}

__global__ __launch_bounds__(256) void dedup_resolve_kernel(DedupParams P) {
    const int locus = P.chunk_locus[blockIdx.x];
    const int64_t lo = P.locus_offsets[locus], hi = P.locus_offsets[locus + 1];
    const int64_t base = lo + (int64_t)P.chunk_index[blockIdx.x] * 1024;
    const bool on = P.on[locus] != 0;
    const unsigned long long size = 2ull * (unsigned long long)(hi - lo);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int64_t c = base + j * 256 + threadIdx.x;
        if (c >= hi) continue;
        // (dup_of is written by plain stores on each path, not through one merged value `rep = same ? r : -1` stored after the
        //  branch.  Round 2 saw wrong representatives with the merged form inside this kernel and took it for a compiler
        //  fault; a stand-alone twin of both forms, tools/microbench/dedup_phi_repro.hip, gives right answers for BOTH on
        //  ROCm 7.2, so the claim is unproven -- the two forms are equivalent and this one is kept)
        P.dup_of[c] = -1;
        if (on && P.flag[c] == TPHIP_FLAG_OK) {
            const unsigned long long h = dedup_key(P.hash[c]);
            unsigned long long slot = h % size;
            while (P.tab_key[2 * lo + slot] != h) slot = (slot + 1 == size) ? 0 : slot + 1;   // inserted by dedup_insert_kernel
            const int64_t r = lo + P.tab_val[2 * lo + slot];
            unsigned differ = (r == c) ? 1u : 0u;
            for (int w = 0; w < P.nwords; ++w)
                differ |= P.packed[(int64_t)w * P.ncols_total + c] ^ P.packed[(int64_t)w * P.ncols_total + r];
            if (differ == 0u) {
                P.dup_of[c] = (int32_t)r;
                P.flag[c] = kFlagDuplicate;
            }
        }
    }
}

__global__ __launch_bounds__(256) void dedup_scatter_kernel(DedupParams P) {
    const int locus = P.chunk_locus[blockIdx.x];
    if (!P.on[locus]) return;
    const int64_t lo = P.locus_offsets[locus], hi = P.locus_offsets[locus + 1];
    const int64_t base = lo + (int64_t)P.chunk_index[blockIdx.x] * 1024;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int64_t c = base + j * 256 + threadIdx.x;
        if (c >= hi) continue;
        const int32_t r = P.dup_of[c];
        if (r < 0) continue;
        P.rate[c] = P.rate[r]; P.subst[c] = P.subst[r]; P.lnl[c] = P.lnl[r]; P.flag[c] = P.flag[r];
    }
}

}  // namespace tphip
