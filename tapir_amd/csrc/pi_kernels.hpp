// pi_kernels.hpp -- column classification / compaction and the PI-table kernels.
//
// classify_kernel + compact_kernel   (HBM-bound byte work: ntaxa bytes read per column)
//     replace HyPhy's unique-pattern bookkeeping (bf:1033-1044) and tapir's informative-site count
//     (tapir/compute.py:96-106): per column the number of plain A/C/G/T cells, and whether the likelihood
//     has a closed-form maximiser (all resolved taxa identical -> s = 0; <= 1 resolved taxon -> flat).
// pi_partial_kernel + pi_reduce_kernel
//     replace bin/tapir_compute.py:114-122: pi = get_townsend_pi(time_vector, rates) is never materialised;
//     the (T x S) matrix is reduced over S on the fly into per-locus rows
//     [net PI(t=0..T-1) | PI at --times | sum(integral) per interval | sum(error) per interval].
//     Summation order is fixed (1024-column blocks, lane-strided, then block order) so a locus gives the
//     same bits no matter how many GPUs the batch is sharded over.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include "gtr_model.hpp"
#include "tree_program.hpp"
#include "quadpack_device.hpp"
#include "tphip.h"

namespace tphip {

constexpr int kPiBlock = 256;
constexpr int kPiColsPerThread = 4;
constexpr int kPiChunk = kPiBlock * kPiColsPerThread;  // 1024 columns per workgroup
constexpr int kTimeTile = 16;

struct ClassifyParams {
    const uint8_t* states;
    int64_t ncols_total;
    int32_t ntaxa;
    const LocusModel* models;
    const int64_t* locus_offsets;
    const int32_t* chunk_locus;   // PI chunks (1024 columns, one locus each)
    const int32_t* chunk_index;
    double* rate;
    double* subst;
    double* lnl;
    uint8_t* flag;
    int32_t* nres;
    double chrono_length;
    const TreeOp* ops;            // the tree program: tips are visited in its order, and its PUSH / POP_MUL structure
    int32_t nops;                 // drives the parsimony pass that seeds the optimiser (ColumnScan::fitch_*)
    uint32_t* packed;             // [ceil(ntaxa/8)][ncols_total] out: 8 four-bit masks per word, program tip order
};

// Per-column bookkeeping shared by the two load paths of classify_kernel.
struct ColumnScan {
    unsigned uni = 0, word = 0;
    int resolved = 0, informative = 0;
    // Fitch parsimony along the tree program: `set` is the state set of the subtree in the accumulator, `stk` the
    // sets of the parked siblings (4 bits per level), `changes` the minimum number of substitutions so far.
    unsigned set = 15u, changes = 0;
    unsigned long long stk = 0;
    __device__ __forceinline__ void fitch_join(unsigned x) {
        const unsigned both = set & x;
        changes += (both == 0u);
        set = both ? both : (set | x);
    }
    __device__ __forceinline__ void fitch_push() { stk = (stk << 4) | set; set = 15u; }
    __device__ __forceinline__ void fitch_pop() { const unsigned x = (unsigned)(stk & 15ull); stk >>= 4; fitch_join(x); }
    __device__ __forceinline__ void tip(unsigned m, int k) {
        m &= 15u;
        m = m ? m : 15u;
        const bool res = (m != 15u);
        uni |= res ? m : 0u;
        resolved += res;
        informative += (__popc(m) == 1);
        word |= m << (4 * (k & 7));
        fitch_join(m);
    }
};

constexpr double kStartA = 0.30, kStartDensityCap = 0.28;

// Where the optimiser starts for a column that needs it: the rate at which the tree would carry the column's
// parsimony count, s0 = changes / (kappa * tree length * fraction of taxa present).  HyPhy starts every column at
// siteRate = 1 (bf:1050), typically e^3 away from the optimum; from s0 (median error 10 %) the same maximum is reached in
// 2.8 instead of 4.4 evaluations on the C3 shape.  Passed to site_rate_kernel through the column's `rate` slot.
__device__ __forceinline__ double start_log_rate(const ClassifyParams& P, const LocusModel* __restrict__ M, const ColumnScan& c) {
    const double len = M->kappa * P.chrono_length * ((double)(c.resolved > 0 ? c.resolved : 1) / (double)P.ntaxa);
    double m = (double)(c.changes > 0u ? c.changes : 1u);
    // parsimony undercounts where changes are dense: with p = changes per branch among the taxa present, the count is
    // stretched to B * (-a ln(1 - p/a)), a = 0.30 (the shape of a multiple-hit correction; a calibrated on the
    // synthetic shapes, where it centres the start on the optimum for p up to 0.28 and saves another 6-12 % of the
    // evaluations; only the starting point depends on it)
    const double B = (double)(2 * c.resolved - 3 > 1 ? 2 * c.resolved - 3 : 1);
    const double pden = fmin(m / B, kStartDensityCap);
    m = B * (-kStartA * log(1.0 - pden / kStartA));
    double u0 = (len > 0.0) ? log(m / len) : 0.0;
    u0 = (u0 == u0) ? u0 : 0.0;
    return fmin(fmax(u0, -20.0), 8.0);
}

__device__ __forceinline__ void classify_finish(const ClassifyParams& P, const LocusModel* __restrict__ M, int64_t col,
                                                const ColumnScan& c, uint8_t& flg_out) {
    uint8_t flg = TPHIP_FLAG_OK;  // provisional: site_rate_kernel will overwrite
    if (c.resolved <= 1) flg = TPHIP_FLAG_FLAT;
    else if (__popc(c.uni) == 1) flg = TPHIP_FLAG_ZERO;
    flg_out = flg;
    if (flg != TPHIP_FLAG_OK) {
        // L = sum over the states allowed by the one informative mask of pi_x (1 if nothing is resolved)
        double L = 0;
        if (c.uni == 0) L = 1.0;
        else {
#pragma unroll
            for (int x = 0; x < 4; ++x) L += ((c.uni >> x) & 1u) ? M->pi[x] : 0.0;
        }
        const double s = (flg == TPHIP_FLAG_FLAT) ? 1.0 : 0.0;  // flat: siteRate keeps its start value (bf:1050)
        const double r = s * M->kappa;
        P.rate[col] = r;
        P.subst[col] = r * P.chrono_length;
        P.lnl[col] = log(L);
    } else {
        P.rate[col] = start_log_rate(P, M, c);   // u0 for site_rate_kernel, which overwrites it with the answer
    }
}

// HBM-bound byte kernel.  Each thread owns 4 CONSECUTIVE columns: when the row addresses are 4-byte aligned
// (alignment of the states pointer, of ncols_total and of the locus offset: true for every BASELINE shape) a taxon
// row is read as one dword per lane (256 B per wave instruction instead of 64), the packed tip words and the
// counts go out as one 16-byte store per lane and the four flags as one dword.  Unaligned batches fall back to
// byte accesses with the same thread-to-column map.  Tips are visited in tree-program order so that the same
// pass emits the packed words site_rate_kernel keeps in registers.
__global__ __launch_bounds__(kPiBlock) void classify_kernel(ClassifyParams P) {
    const int locus = P.chunk_locus[blockIdx.x];
    const int64_t lo = P.locus_offsets[locus], hi = P.locus_offsets[locus + 1];
    const LocusModel* __restrict__ M = P.models + locus;
    const int64_t c0 = lo + (int64_t)P.chunk_index[blockIdx.x] * kPiChunk + 4 * (int64_t)threadIdx.x;
    if (c0 >= hi) return;
    const bool full = (c0 + 4 <= hi);
    const bool aligned = full && ((reinterpret_cast<uintptr_t>(P.states) & 3u) == 0) && ((P.ncols_total & 3) == 0) &&
                         ((c0 & 3) == 0);
    ColumnScan c[4];
    if (aligned) {
        int k = 0;
        for (int ip = 0; ip < P.nops; ++ip) {
            const int code = P.ops[ip].code;   // uniform -> scalar loads
            if (code == OP_PUSH) {
#pragma unroll
                for (int j = 0; j < 4; ++j) c[j].fitch_push();
                continue;
            }
            if (code == OP_POP_MUL) {
#pragma unroll
                for (int j = 0; j < 4; ++j) c[j].fitch_pop();
                continue;
            }
            if (code == OP_BRANCH) continue;
            const int t = P.ops[ip].taxon;
            const uint32_t v = *reinterpret_cast<const uint32_t*>(P.states + (int64_t)t * P.ncols_total + c0);
#pragma unroll
            for (int j = 0; j < 4; ++j) c[j].tip((v >> (8 * j)) & 0xffu, k);
            if ((k & 7) == 7 || k == P.ntaxa - 1) {
                uint4 w = make_uint4(c[0].word, c[1].word, c[2].word, c[3].word);
                *reinterpret_cast<uint4*>(P.packed + (int64_t)(k >> 3) * P.ncols_total + c0) = w;
#pragma unroll
                for (int j = 0; j < 4; ++j) c[j].word = 0;
            }
            ++k;
        }
        uint8_t f[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) classify_finish(P, M, c0 + j, c[j], f[j]);
        *reinterpret_cast<int4*>(P.nres + c0) = make_int4(c[0].informative, c[1].informative, c[2].informative, c[3].informative);
        *reinterpret_cast<uint32_t*>(P.flag + c0) = (uint32_t)f[0] | ((uint32_t)f[1] << 8) | ((uint32_t)f[2] << 16) | ((uint32_t)f[3] << 24);
    } else {
        const int n = full ? 4 : (int)(hi - c0);
        int k = 0;
        for (int ip = 0; ip < P.nops; ++ip) {
            const int code = P.ops[ip].code;
            if (code == OP_PUSH) {
#pragma unroll
                for (int j = 0; j < 4; ++j) c[j].fitch_push();
                continue;
            }
            if (code == OP_POP_MUL) {
#pragma unroll
                for (int j = 0; j < 4; ++j) c[j].fitch_pop();
                continue;
            }
            if (code == OP_BRANCH) continue;
            const int t = P.ops[ip].taxon;
            const uint8_t* row = P.states + (int64_t)t * P.ncols_total + c0;
#pragma unroll
            for (int j = 0; j < 4; ++j) if (j < n) c[j].tip(row[j], k);
            if ((k & 7) == 7 || k == P.ntaxa - 1) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if (j < n) P.packed[(int64_t)(k >> 3) * P.ncols_total + c0 + j] = c[j].word;
                    c[j].word = 0;
                }
            }
            ++k;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (j < n) {
                uint8_t f;
                classify_finish(P, M, c0 + j, c[j], f);
                P.nres[c0 + j] = c[j].informative;
                P.flag[c0 + j] = f;
            }
        }
    }
}

// One workgroup per locus: stable compaction of the columns that still need the optimiser.
__global__ __launch_bounds__(256) void compact_kernel(const uint8_t* __restrict__ flag, const int64_t* __restrict__ locus_offsets,
                                                      int32_t* __restrict__ work_cols, int32_t* __restrict__ work_count) {
    __shared__ int wave_tot[4];
    __shared__ int running;
    const int locus = blockIdx.x;
    const int64_t lo = locus_offsets[locus], hi = locus_offsets[locus + 1];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) running = 0;
    __syncthreads();
    for (int64_t base = lo; base < hi; base += 256) {
        const int64_t col = base + threadIdx.x;
        const bool want = (col < hi) && (flag[col] == TPHIP_FLAG_OK);
        const unsigned long long bal = __ballot(want);
        const int before = __popcll(bal & ((1ull << lane) - 1ull));
        if (lane == 0) wave_tot[wave] = __popcll(bal);
        __syncthreads();
        int off = running;
        for (int w = 0; w < wave; ++w) off += wave_tot[w];
        if (want) work_cols[lo + off + before] = (int32_t)col;
        __syncthreads();
        if (threadIdx.x == 0) running += wave_tot[0] + wave_tot[1] + wave_tot[2] + wave_tot[3];
        __syncthreads();
    }
    if (threadIdx.x == 0) work_count[locus] = running;
}

struct PiParams {
    const double* rates;       // raw stage-1 rates (kappa * s) or user-supplied rates
    const int32_t* nres;       // may be null (no cull)
    const int64_t* locus_offsets;
    const int32_t* chunk_locus;
    const int32_t* chunk_index;
    int32_t T;
    const int32_t* intervals;  // [n_i][2]
    int32_t n_i;
    int32_t integ_mode;
    double correction;
    int32_t threshold;
    double round_scale;        // 10^decimals, or 0 for no rounding
    double* partial;           // [nchunks][T + 2 n_i]
};

// rate as tapir sees it after the JSON round trip, the /correction and the cull
// (bf:1093-1095 Format(x,0,4); tapir/compute.py:38-39; tapir/compute.py:96-110)
__device__ __forceinline__ double finalize_rate(const PiParams& P, int64_t col) {
    double r = P.rates[col];
    if (P.round_scale > 0.0) r = rint(r * P.round_scale) / P.round_scale;
    r = r / P.correction;
    if (P.nres && P.nres[col] < P.threshold) r = __longlong_as_double(0x7ff8000000000000ll);
    return r;
}

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

__global__ __launch_bounds__(kPiBlock) void pi_partial_kernel(PiParams P) {
    __shared__ double red[kTimeTile][4];
    // per-tile transpose buffer: [time][16 segments of 16 threads, each padded to 17 doubles]
    __shared__ double tile[kTimeTile][16 * 17];
    const int chunk = blockIdx.x;
    const int locus = P.chunk_locus[chunk];
    const int64_t lo = P.locus_offsets[locus], hi = P.locus_offsets[locus + 1];
    const int64_t base = lo + (int64_t)P.chunk_index[chunk] * kPiChunk;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    double r[kPiColsPerThread];
    bool ok[kPiColsPerThread];
#pragma unroll
    for (int j = 0; j < kPiColsPerThread; ++j) {
        const int64_t col = base + j * kPiBlock + threadIdx.x;
        r[j] = (col < hi) ? finalize_rate(P, col) : __longlong_as_double(0x7ff8000000000000ll);
        ok[j] = isfinite(r[j]);  // nansum skips NaN (bin/tapir_compute.py:119); quad only sees finite rates (:122)
    }
    const int Wp = P.T + 2 * P.n_i;
    double* out = P.partial + (size_t)chunk * Wp;
    // ---- net PI(t), t = 0..T-1, in tiles of kTimeTile time points
    for (int t0 = 0; t0 < P.T; t0 += kTimeTile) {
        double acc[kTimeTile];
#pragma unroll
        for (int i = 0; i < kTimeTile; ++i) acc[i] = 0.0;
#pragma unroll
        for (int j = 0; j < kPiColsPerThread; ++j) {
            if (!ok[j]) continue;
            const double c = 16.0 * (r[j] * r[j]);
            const double q = exp(-(4.0 * r[j]));
            double p = exp(-(4.0 * r[j] * (double)t0));
#pragma unroll
            for (int i = 0; i < kTimeTile; ++i) {
                acc[i] = fma(c * (double)(t0 + i), p, acc[i]);
                p *= q;
            }
        }
        // Sum over the 256 threads WITHOUT 16 butterfly reductions (96 cross-lane FP64 exchanges per wave cost ten
        // times the arithmetic above): transpose through LDS, each thread then adds one 16-value segment of one
        // time point in a fixed order, and a 4-step exchange inside 16-lane groups finishes the sum.
        {
            const int slot = (threadIdx.x >> 4) * 17 + (threadIdx.x & 15);
#pragma unroll
            for (int i = 0; i < kTimeTile; ++i) tile[i][slot] = acc[i];
        }
        __syncthreads();
        {
            const int i = threadIdx.x >> 4, seg = threadIdx.x & 15;
            const double* src = &tile[i][seg * 17];
            double sum = 0.0;
#pragma unroll
            for (int k = 0; k < 16; ++k) sum += src[k];
            sum += __shfl_xor(sum, 8);
            sum += __shfl_xor(sum, 4);
            sum += __shfl_xor(sum, 2);
            sum += __shfl_xor(sum, 1);
            if (seg == 0 && t0 + i < P.T) out[t0 + i] = sum;
        }
        __syncthreads();
    }
    // ---- interval integrals
    for (int k = 0; k < P.n_i; ++k) {
        const double a = (double)P.intervals[2 * k], b = (double)P.intervals[2 * k + 1];
        double si = 0.0, se = 0.0;
#pragma unroll 1
        for (int j = 0; j < kPiColsPerThread; ++j) {
            if (!ok[j]) continue;
            double res, err = 0.0;
            // (measured: the table-driven exp of fast_exp.hpp makes this kernel 1.65x SLOWER -- 84 scattered LDS
            //  reads per column from one table shared by 256 threads -- so the quadrature keeps the library exp)
            if (P.integ_mode == TPHIP_INTEG_QUADPACK) quad_townsend<false>(a, b, r[j], nullptr, res, err);
            else res = integral_closed(a, b, r[j]);
            si += res;
            se += err;
        }
        si = wave_sum(si);
        se = wave_sum(se);
        if (lane == 0) { red[0][wave] = si; red[1][wave] = se; }
        __syncthreads();
        if (threadIdx.x < 2)
            out[P.T + threadIdx.x * P.n_i + k] =
                ((red[threadIdx.x][0] + red[threadIdx.x][1]) + red[threadIdx.x][2]) + red[threadIdx.x][3];
        __syncthreads();
    }
}

// tables[l] = [net(T) | disc(n_t) | integral(n_i) | error(n_i)], summing the locus' chunks in order.
__global__ void pi_reduce_kernel(const double* __restrict__ partial, const int64_t* __restrict__ locus_chunk_offsets,
                                 int32_t T, const int32_t* __restrict__ times, int32_t n_t, int32_t n_i,
                                 double* __restrict__ tables) {
    const int locus = blockIdx.x;
    const int64_t c0 = locus_chunk_offsets[locus], c1 = locus_chunk_offsets[locus + 1];
    const int Wp = T + 2 * n_i, W = T + n_t + 2 * n_i;
    double* row = tables + (size_t)locus * W;
    for (int w = threadIdx.x; w < Wp + n_t; w += blockDim.x) {
        // w < Wp: a partial column; else a --times entry, which is net[times[k]] (tapir/compute.py:76-79)
        const int src = (w < Wp) ? w : times[w - Wp];
        double s = 0.0;
        for (int64_t c = c0; c < c1; ++c) s += partial[(size_t)c * Wp + src];
        const int dst = (w < T) ? w : (w < Wp ? w + n_t : T + (w - Wp));
        row[dst] = s;
    }
}

// tapir/compute.py:46-48 as a dense (n_times, n) matrix: out[k*n + i] = 16 r_i^2 t_k exp(-4 r_i t_k).
// Pure streaming: 8 B read per site, 8*n_times B written; consecutive lanes write consecutive doubles.
__global__ void townsend_dense_kernel(const double* __restrict__ rates, int64_t n, const double* __restrict__ times,
                                      int32_t n_times, double* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double r = rates[i];
    for (int k = 0; k < n_times; ++k) out[(size_t)k * n + i] = townsend_pi<false>(times[k], r, nullptr);
}

// tapir/compute.py:50-52 vectorised over sites (what numpy.vectorize(get_integral_over_times) returns)
__global__ void quad_sites_kernel(const double* __restrict__ rates, int64_t n, double a, double b, int32_t integ_mode,
                                  double* __restrict__ integral, double* __restrict__ abserr) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double r = rates[i];
    double res = __longlong_as_double(0x7ff8000000000000ll), err = res;
    if (isfinite(r)) {
        err = 0.0;
        if (integ_mode == TPHIP_INTEG_QUADPACK) quad_townsend<false>(a, b, r, nullptr, res, err);
        else res = integral_closed(a, b, r);
    }
    integral[i] = res;
    abserr[i] = err;
}

// per-locus histogram of the 16 state masks (HarvestFrequencies, bf:968)
__global__ __launch_bounds__(256) void state_histogram_kernel(const uint8_t* __restrict__ states, int64_t ncols_total,
                                                              int32_t ntaxa, const int64_t* __restrict__ locus_offsets,
                                                              unsigned long long* __restrict__ hist) {
    __shared__ unsigned int h[16];
    const int locus = blockIdx.x;
    const int64_t lo = locus_offsets[locus], hi = locus_offsets[locus + 1];
    if (threadIdx.x < 16) h[threadIdx.x] = 0;
    __syncthreads();
    unsigned int mine[16];
#pragma unroll
    for (int m = 0; m < 16; ++m) mine[m] = 0;
    for (int t = 0; t < ntaxa; ++t) {
        const uint8_t* row = states + (int64_t)t * ncols_total;
        for (int64_t c = lo + threadIdx.x; c < hi; c += blockDim.x) {
            unsigned m = row[c] & 15u;
            m = m ? m : 15u;
#pragma unroll
            for (int k = 0; k < 16; ++k) mine[k] += (m == (unsigned)k);
        }
    }
#pragma unroll
    for (int m = 0; m < 16; ++m) {
        unsigned v = mine[m];
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
        if ((threadIdx.x & 63) == 0) atomicAdd(&h[m], v);
    }
    __syncthreads();
    if (threadIdx.x < 16) hist[(size_t)locus * 16 + threadIdx.x] = h[threadIdx.x];
}

}  // namespace tphip
