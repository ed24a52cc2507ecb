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
#include "site_rate_params.hpp"
#include "tphip.h"

namespace tphip {

constexpr int kPiBlock = 256;
constexpr int kPiColsPerThread = 4;
constexpr int kPiChunk = kPiBlock * kPiColsPerThread;  // 1024 columns per workgroup
constexpr int kTimeTile = 16;
constexpr int kIvTile = 8;      // intervals per register tile of pi_partial_kernel (2 * kIvTile <= kTimeTile rows of `red`)

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
    const int32_t* tip_taxon;     // [ntaxa] alignment row of the k-th tip of the program
    double start_scale;           // 1: start at the parsimony rate; 0: at HyPhy's siteRate = 1 (bf:1050), TPHIP_START_*
    uint64_t* hash;               // [ncols_total] out (may be null): hash of the column's packed words, for the
                                  // per-pattern de-duplication of the site-rate stage (pattern_kernels.hpp)
};

// Scalar loads of wave-uniform table entries.  classify_kernel also stores, so through plain pointers the compiler
// cannot prove the tables invariant and fetches them with vector loads (one memory round trip per op, per lane).
__device__ __forceinline__ int32_t scalar_load_i32(const int32_t* p) {   // p must be wave-uniform
    int32_t r;
    asm volatile("s_load_dword %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(r) : "s"(p));
    return r;
}
typedef int SI8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ SI8 scalar_load_i32x8(const int32_t* p) {     // p wave-uniform and 32-byte aligned
    SI8 r;
    asm volatile("s_load_dwordx8 %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(r) : "s"(p));
    return r;
}

// Per-column bookkeeping of classify_kernel, as its finishing code reads it.
struct ColumnScan {
    unsigned uni = 0;
    int resolved = 0, informative = 0;
    unsigned changes = 0;   // Fitch parsimony: minimum number of substitutions on the tree
    int base[3] = {0, 0, 0};   // plain A, C, G cells (T = informative - the three): the column's own exit rate, start_log_rate
};

// Four consecutive columns at once: a thread's four state bytes of one taxon arrive as one dword, and every quantity
// classify_kernel needs per tip is a per-byte operation on 4-bit masks (SWAR): 10 integer instructions per column and
// tip instead of 30 with one ColumnScan per column, which made this byte kernel VALU-bound at 0.48 ms on C3 (its
// 575 MB move in ~0.15 ms).
//   * `set4` / `stk`: Fitch parsimony along the tree program -- the state set of the subtree in the accumulator (one
//     byte per column) and the sets of the parked siblings (4 bits per level and column);
//   * counters (resolved taxa, plain A/C/G/T cells, Fitch changes) accumulate one byte per column and are flushed
//     into 32-bit totals every 64 tips.
struct ColumnScan4 {
    static constexpr uint32_t k01 = 0x01010101u, k0f = 0x0f0f0f0fu;
    uint32_t uni4 = 0, set4 = k0f;
    uint32_t res8 = 0, inf8 = 0, chg8 = 0;            // byte counters since the last flush
    uint32_t a8 = 0, c8 = 0, g8 = 0;                  // ... of the plain A / C / G cells
    int base[3][4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    uint32_t word[4] = {0, 0, 0, 0};
    uint64_t hsh[4] = {0x243F6A8885A308D3ull, 0x243F6A8885A308D3ull, 0x243F6A8885A308D3ull, 0x243F6A8885A308D3ull};
    int resolved[4] = {0, 0, 0, 0}, informative[4] = {0, 0, 0, 0};
    unsigned changes[4] = {0, 0, 0, 0};
    unsigned long long stk[4] = {0, 0, 0, 0};
    // 0x01 in every byte whose low nibble (the only bits set) is nonzero / equals 15
    static __device__ __forceinline__ uint32_t nonzero(uint32_t x) { return ((x + k0f) >> 4) & k01; }
    static __device__ __forceinline__ uint32_t is15(uint32_t x) { return ((x + k01) >> 4) & k01; }
    static __device__ __forceinline__ uint32_t spread(uint32_t b) { return (b << 4) - b; }   // 0x01 -> 0x0f per byte
    __device__ __forceinline__ void flush() {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            resolved[j] += (int)((res8 >> (8 * j)) & 0xffu);
            informative[j] += (int)((inf8 >> (8 * j)) & 0xffu);
            changes[j] += (chg8 >> (8 * j)) & 0xffu;
            base[0][j] += (int)((a8 >> (8 * j)) & 0xffu);
            base[1][j] += (int)((c8 >> (8 * j)) & 0xffu);
            base[2][j] += (int)((g8 >> (8 * j)) & 0xffu);
        }
        res8 = inf8 = chg8 = 0;
        a8 = c8 = g8 = 0;
    }
    __device__ __forceinline__ void fitch_join(uint32_t x4) {
        const uint32_t both = set4 & x4;
        const uint32_t empty = nonzero(both) ^ k01;
        chg8 += empty;
        set4 = both | ((set4 | x4) & spread(empty));
    }
    __device__ __forceinline__ void fitch_push() {
#pragma unroll
        for (int j = 0; j < 4; ++j) stk[j] = (stk[j] << 4) | ((set4 >> (8 * j)) & 15u);
        set4 = k0f;
    }
    __device__ __forceinline__ void fitch_pop() {
        uint32_t x4 = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) { x4 |= (uint32_t)(stk[j] & 15ull) << (8 * j); stk[j] >>= 4; }
        fitch_join(x4);
    }
    // v: the state bytes of tip number k (program order) for the four columns
    __device__ __forceinline__ void tip(uint32_t v, int k) {
        uint32_t m4 = v & k0f;
        m4 |= spread(nonzero(m4) ^ k01);              // code 0 (nothing allowed) reads as "anything": 15
        const uint32_t res = is15(m4) ^ k01;          // resolved: not a gap / ? / N
        uni4 |= m4 & spread(res);
        res8 += res;
        const uint32_t single = nonzero(m4 & (m4 - k01)) ^ k01;   // exactly one bit set (m4 >= 1 in every byte: no borrow)
        inf8 += single;
        a8 += single & m4;                            // ... and it is bit 0 / 1 / 2 (the neighbour byte's bit that a shift
        c8 += single & (m4 >> 1);                     // brings into bit 7 is masked away by `single`)
        g8 += single & (m4 >> 2);
#pragma unroll
        for (int j = 0; j < 4; ++j) word[j] |= ((m4 >> (8 * j)) & 15u) << (4 * (k & 7));
        fitch_join(m4);
        if ((k & 63) == 63) flush();   // <= 64 tip joins + <= 64 + 16 sibling joins per window: no byte overflows
    }
    __device__ __forceinline__ ColumnScan column(int j) const {   // after flush()
        ColumnScan c;
        c.uni = (uni4 >> (8 * j)) & 15u;
        c.resolved = resolved[j];
        c.informative = informative[j];
        c.changes = changes[j];
        c.base[0] = base[0][j]; c.base[1] = base[1][j]; c.base[2] = base[2][j];
        return c;
    }
};

constexpr double kStartA = 0.25, kStartDensityCap = 0.235;   // (0.30 / 0.28 before the column's own exit rate went into the start)

// Where the optimiser starts for a column that needs it: the rate at which the tree would carry the column's
// parsimony count, s0 = changes / (kappa * tree length * fraction of taxa present).  HyPhy starts every column at
// siteRate = 1 (bf:1050), typically e^3 away from the optimum; from s0 (median error 10 %) the same maximum is reached in
// 2.8 instead of 4.4 evaluations on the C3 shape.  Passed to site_rate_kernel through the column's `rate` slot.
// -Q_xx of the locus' model for x = A, C, G, T: -sum_k U[x][k] lam_k U^-1[k][x] (lam_0 = 0); once per thread, the four columns
// of a thread belong to one locus
__device__ __forceinline__ void exit_rates(const LocusModel* __restrict__ M, double* ex) {
#pragma unroll
    for (int x = 0; x < 4; ++x) {
        double qxx = 0.0;
#pragma unroll
        for (int k = 0; k < 3; ++k) qxx = fma(M->U[x * 3 + k] * M->lam[k], M->Ui[k * 4 + x], qxx);
        ex[x] = -qxx;
    }
}

__device__ __forceinline__ double start_log_rate(const ClassifyParams& P, const LocusModel* __restrict__ M, const ColumnScan& c,
                                                 const double* ex) {
    // Rate at which THIS column's states are left: sum_x p_x (-Q_xx) over its plain cells instead of the stationary mean
    // kappa = sum_x pi_x (-Q_xx).  A column of mostly fast-leaving bases reaches its parsimony count at a lower site rate; with
    // the column's own exit rate the start lands 2-3 x closer (64 taxa: rms miss 0.20 -> 0.105 log-units, 256 taxa: 0.36 ->
    // 0.13; same rule as the oracle).  ex[x] = -Q_xx (exit_rates).
    double exit_rate = M->kappa;
    if (c.informative > 0) {
        const int cnt[4] = {c.base[0], c.base[1], c.base[2], c.informative - c.base[0] - c.base[1] - c.base[2]};
        double acc = 0.0;
#pragma unroll
        for (int x = 0; x < 4; ++x) acc = fma((double)cnt[x], ex[x], acc);
        if (acc > 0.0) exit_rate = acc / (double)c.informative;
    }
    const double len = exit_rate * P.chrono_length * ((double)(c.resolved > 0 ? c.resolved : 1) / (double)P.ntaxa);
    double m = (double)(c.changes > 0u ? c.changes : 1u);
    // parsimony undercounts where changes are dense: with p = changes per branch among the taxa present, the count is
    // stretched to B * (-a ln(1 - p/a)), a = kStartA (the shape of a multiple-hit correction; a calibrated on the
    // synthetic shapes, where it centres the start on the optimum for p up to kStartDensityCap and saves another 6-12 %
    // of the evaluations; only the starting point depends on it)
    const double B = (double)(2 * c.resolved - 3 > 1 ? 2 * c.resolved - 3 : 1);
    const double pden = fmin(m / B, kStartDensityCap);
    m = B * (-kStartA * log(1.0 - pden / kStartA));
    double u0 = (len > 0.0) ? log(m / len) : 0.0;
    u0 = (u0 == u0) ? u0 : 0.0;
    return fmin(fmax(u0, -20.0), 8.0);
}

__device__ __forceinline__ void classify_finish(const ClassifyParams& P, const LocusModel* __restrict__ M, int64_t col,
                                                const ColumnScan& c, const double* ex, uint8_t& flg_out) {
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
        // u0 for site_rate_kernel, which overwrites it with the answer: the parsimony start, or u = log 1 = 0 when the
        // plan asks for HyPhy's start value (start_scale = 0; a multiplication, not a branch: + 0.0 turns -0.0 into 0.0)
        P.rate[col] = start_log_rate(P, M, c, ex) * P.start_scale + 0.0;
        // the column's parsimony length for the optimiser's first step (site_rate_kernel overwrites the slot with the answer);
        // 0 = none: HyPhy's start value asks for the plain step from there, and on small trees an evaluation is so cheap that
        // the extra logarithm costs more than the saved evaluations (C2, 16 taxa: +4 % time with it; C3 -2 %, C5 -4 %)
        P.subst[col] = (P.ntaxa >= kFirstStepMinTaxa) ? (double)c.changes * P.start_scale : 0.0;
        // ... and in the `lnl` slot (also overwritten with the answer) whether the column is likely to need many evaluations:
        // a parsimony count of 30 % or more of the resolved taxa.  Small batches (slice and mixed-loci modes of site_rate_kernel) last as
        // long as their slowest wave; a slow column (one in a hundred needs 9-17 evaluations from siteRate = 1, all of them
        // in this class: tools/debug note in DESIGN section 8 r3) taken late keeps its wave alive alone, so those waves
        // take the marked columns first (site_rate_kernel, small-batch mode).  The order changes no column's result.
        P.lnl[col] = (10u * c.changes >= 3u * (c.resolved > 1u ? c.resolved - 1u : 1u)) ? 1.0 : 0.0;
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
    const int n = full ? 4 : (int)(hi - c0);
    const bool fast = __all(aligned);   // wave-uniform: only the last wave of a locus can have a ragged thread
    ColumnScan4 c;
    uint32_t buf[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    int k = 0;
    for (int ip = 0; ip < P.nops; ++ip) {
        const int code = scalar_load_i32(&P.ops[ip].code);
        if (code == OP_PUSH) { c.fitch_push(); continue; }
        if (code == OP_POP_MUL) { c.fitch_pop(); continue; }
        if (code == OP_BRANCH) continue;
        // The state dwords of eight tips are requested together when the first of them comes up: fetched one tip
        // at a time (address from the op just read, value needed at once) every tip cost a full memory round trip
        // and the kernel ran at 1.2 TB/s whatever its arithmetic.  (tip_taxon is padded to a multiple of 8.)
        if ((k & 7) == 0) {
            const SI8 tx = scalar_load_i32x8(P.tip_taxon + k);
            const int t8[8] = {tx.s0, tx.s1, tx.s2, tx.s3, tx.s4, tx.s5, tx.s6, tx.s7};
            if (fast) {
#pragma unroll
                for (int i = 0; i < 8; ++i)
                    buf[i] = *reinterpret_cast<const uint32_t*>(P.states + (int64_t)t8[i] * P.ncols_total + c0);
            } else {
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const uint8_t* row = P.states + (int64_t)t8[i] * P.ncols_total + c0;
                    uint32_t w = 0;
#pragma unroll
                    for (int j = 0; j < 4; ++j) if (j < n) w |= (uint32_t)row[j] << (8 * j);   // missing columns read as gaps
                    buf[i] = w;
                }
            }
        }
        uint32_t v = buf[0];
#pragma unroll
        for (int i = 1; i < 8; ++i) v = ((k & 7) == i) ? buf[i] : v;   // k is wave-uniform: scalar compares
        c.tip(v, k);
        if ((k & 7) == 7 || k == P.ntaxa - 1) {
            if (fast) {
                *reinterpret_cast<uint4*>(P.packed + (int64_t)(k >> 3) * P.ncols_total + c0) = make_uint4(c.word[0], c.word[1], c.word[2], c.word[3]);
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) if (j < n) P.packed[(int64_t)(k >> 3) * P.ncols_total + c0 + j] = c.word[j];
            }
            if (P.hash) {   // wave-uniform
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    uint64_t h = (c.hsh[j] ^ c.word[j]) * 0x9E3779B97F4A7C15ull;
                    c.hsh[j] = h ^ (h >> 29);
                }
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) c.word[j] = 0;
        }
        ++k;
    }
    c.flush();
    uint8_t f[4] = {0, 0, 0, 0};
    double ex[4];
    exit_rates(M, ex);
#pragma unroll
    for (int j = 0; j < 4; ++j) if (j < n) classify_finish(P, M, c0 + j, c.column(j), ex, f[j]);
    if (P.hash) {
#pragma unroll
        for (int j = 0; j < 4; ++j) if (j < n) P.hash[c0 + j] = c.hsh[j] ^ (c.hsh[j] >> 32);
    }
    if (fast) {
        *reinterpret_cast<int4*>(P.nres + c0) = make_int4(c.informative[0], c.informative[1], c.informative[2], c.informative[3]);
        *reinterpret_cast<uint32_t*>(P.flag + c0) = (uint32_t)f[0] | ((uint32_t)f[1] << 8) | ((uint32_t)f[2] << 16) | ((uint32_t)f[3] << 24);
    } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (j < n) { P.nres[c0 + j] = c.informative[j]; P.flag[c0 + j] = f[j]; }
        }
    }
}

// One workgroup per locus: stable compaction of the columns that still need the optimiser.  The loop is a chain of
// barriers, so its time is the number of rounds: long loci get 1024 threads (C3: 49 rounds per locus instead of 196),
// short ones 256.
template <int kCompactBlock>
__global__ __launch_bounds__(kCompactBlock) void compact_kernel(const uint8_t* __restrict__ flag, const int64_t* __restrict__ locus_offsets,
                                                                int32_t* __restrict__ work_cols, int32_t* __restrict__ work_count) {
    constexpr int kWaves = kCompactBlock / 64;
    __shared__ int wave_tot[kWaves];
    __shared__ int running;
    const int locus = blockIdx.x;
    const int64_t lo = locus_offsets[locus], hi = locus_offsets[locus + 1];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) running = 0;
    __syncthreads();
    for (int64_t base = lo; base < hi; base += kCompactBlock) {
        const int64_t col = base + threadIdx.x;
        const bool want = (col < hi) && (flag[col] == TPHIP_FLAG_OK);
        const unsigned long long bal = __ballot(want);
        const int before = __popcll(bal & ((1ull << lane) - 1ull));
        if (lane == 0) wave_tot[wave] = __popcll(bal);
        __syncthreads();
        int off = running, tot = 0;
#pragma unroll
        for (int w = 0; w < kWaves; ++w) { off += (w < wave) ? wave_tot[w] : 0; tot += wave_tot[w]; }
        if (want) work_cols[lo + off + before] = (int32_t)col;
        __syncthreads();
        if (threadIdx.x == 0) running += tot;
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
// HyPhy writes Format(x,0,4) (bf:1093-1095) and tapir parses the text back: the double nearest to the decimal that
// printf-style rounding of the EXACT binary value gives (ties to even).  rint(r * 10^4) / 10^4 is not that: the product
// is itself rounded, so it can land exactly on a half-integer the true product only comes close to (a false tie, decided
// by parity instead of by the discarded part) -- the rate would then differ by 1e-4 from what the .rates file says.
// The FMA recovers the discarded part exactly (r * scale = p + e), and only a half-integer p needs it: otherwise p is
// at least one ulp away from the half and |e| <= ulp/2 cannot carry it across.
// (Contraction must stay off here: fused into fma(r, scale, -n), "p - n" would be the exact product minus n, never
// exactly 0.5 on a false tie, and the correction below would never fire.)
__host__ __device__ __forceinline__ double round_like_printf(double r, double scale) {
#pragma clang fp contract(off)
    const double p = r * scale;
    const double e = fma(r, scale, -p);
    double n = rint(p);
    if (fabs(p - n) == 0.5 && e != 0.0) n = (e > 0.0) ? p + 0.5 : p - 0.5;
    return n / scale;   // correctly rounded quotient of two exact doubles = strtod of the decimal string
}

__device__ __forceinline__ double finalize_rate(const PiParams& P, int64_t col) {
    double r = P.rates[col];
    if (P.round_scale > 0.0) r = round_like_printf(r, P.round_scale);
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
        // nansum skips NaN (bin/tapir_compute.py:119); quad only sees finite rates (:122); and a rate of exactly 0
        // (constant columns: 22-41 % of the synthetic shapes) contributes exactly 0 to every sum, error estimates included
        ok[j] = isfinite(r[j]) && r[j] != 0.0;
    }
    // Pack the chunk's contributing rates to the front (thread-major order, fixed for a given chunk): the slots the
    // culled and constant columns would have occupied are whole waves that skip the body below instead of lanes idling
    // inside it (C2: 290 of a locus' 500 columns contribute -- 5 wave-slots instead of 8).
    {
        __shared__ double packed_rate[kPiChunk];
        __shared__ int wave_count[kPiBlock / 64];
        int c = 0;
#pragma unroll
        for (int j = 0; j < kPiColsPerThread; ++j) c += ok[j] ? 1 : 0;
        int inc = c;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int up = __shfl_up(inc, o);
            if (lane >= o) inc += up;
        }
        if (lane == 63) wave_count[wave] = inc;
        __syncthreads();
        int pos = inc - c, total = 0;
#pragma unroll
        for (int w = 0; w < kPiBlock / 64; ++w) { pos += (w < wave) ? wave_count[w] : 0; total += wave_count[w]; }
#pragma unroll
        for (int j = 0; j < kPiColsPerThread; ++j) if (ok[j]) packed_rate[pos++] = r[j];
        __syncthreads();
#pragma unroll
        for (int j = 0; j < kPiColsPerThread; ++j) {
            const int idx = j * kPiBlock + (int)threadIdx.x;
            ok[j] = idx < total;
            r[j] = ok[j] ? packed_rate[idx] : __longlong_as_double(0x7ff8000000000000ll);
        }
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
    // ---- interval integrals, in tiles of kIvTile intervals: per column the first-panel exponentials of consecutive
    // intervals of equal length are computed once (quadpack_device.hpp: GkFactors).  Per (thread, interval) the columns
    // are still added in the order j = 0..3, then lanes, then waves: the sums keep their fixed order.
    for (int k0 = 0; k0 < P.n_i; k0 += kIvTile) {
        double si[kIvTile], se[kIvTile];
#pragma unroll
        for (int kk = 0; kk < kIvTile; ++kk) { si[kk] = 0.0; se[kk] = 0.0; }
#pragma unroll 1
        for (int j = 0; j < kPiColsPerThread; ++j) {
            if (!ok[j]) continue;
            GkFactors F;
            double have_h = -1.0;        // half-length the factors in F were built for (intervals are wave-uniform)
#pragma unroll
            for (int kk = 0; kk < kIvTile; ++kk) {
                const int k = k0 + kk;
                if (k < P.n_i) {         // wave-uniform
                    const double a = (double)P.intervals[2 * k], b = (double)P.intervals[2 * k + 1];
                    double res, err = 0.0;
                    // (measured: the table-driven exp of fast_exp.hpp makes this kernel 1.65x SLOWER -- scattered LDS reads
                    //  from one table shared by 256 threads -- and its degree-13 polynomial 1.3x: the library exp stays)
                    if (P.integ_mode == TPHIP_INTEG_QUADPACK) {
                        const double hl = 0.5 * (b - a);
                        if (4.0 * r[j] * hl < 30.0) {        // per lane: a huge rate takes the generic panel
                            if (hl != have_h) { gk_factors(r[j], hl, F); have_h = hl; }
                            quad_townsend_factored(a, b, r[j], F, res, err);
                        } else {
                            quad_townsend<false>(a, b, r[j], nullptr, res, err);
                        }
                    } else {
                        res = integral_closed(a, b, r[j]);
                    }
                    si[kk] += res;
                    se[kk] += err;
                }
            }
        }
#pragma unroll
        for (int kk = 0; kk < kIvTile; ++kk) {
            const int k = k0 + kk;
            if (k < P.n_i) {
                const double s1 = wave_sum(si[kk]), s2 = wave_sum(se[kk]);
                if (lane == 0) { red[2 * kk][wave] = s1; red[2 * kk + 1][wave] = s2; }
            }
        }
        __syncthreads();
        if (threadIdx.x < 2 * kIvTile && k0 + (int)(threadIdx.x >> 1) < P.n_i) {
            const int kk = threadIdx.x >> 1, which = threadIdx.x & 1;
            out[P.T + which * P.n_i + k0 + kk] =
                ((red[threadIdx.x][0] + red[threadIdx.x][1]) + red[threadIdx.x][2]) + red[threadIdx.x][3];
        }
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

// parse_site_rates + cull_uninformative_rates as the PI kernels see a rate (tapir/compute.py:38-39, 108-110):
// out[c] = round4(rate[c]) / correction, NaN where nres[c] < threshold (nres may be null: no cull).
__global__ void corrected_rates_kernel(PiParams P, int64_t n, double* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = finalize_rate(P, i);
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
