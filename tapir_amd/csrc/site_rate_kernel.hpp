// site_rate_kernel.hpp -- per-column substitution-rate ML (HyPhy stage 2) on gfx950.
//
// Reference behaviour replaced: tapir/data/models_and_rates.bf:1042-1070 -- for each alignment column
// `Optimize(site_res, siteLikelihood)` over the single scalar siteRate that multiplies every branch length
// of the fixed tree (bf:1003-1013), GTR model (bf:978-1001); HyPhy's start value is 1 (bf:1050), here the column's
// parsimony rate computed by classify_kernel (same maximum, one evaluation fewer).
//
// Mapping to CDNA4
//   * one alignment column per lane, one wavefront per workgroup, all of one locus, so the locus' eigen-system
//     (31 doubles) and the tree program are wave-uniform.  The program arrives by hand-issued scalar loads two ops
//     ahead; the model sits in (lane-replicated) VGPRs, loaded through LDS: 62 SGPRs of model do not fit beside the
//     op pipeline and the pointers, and the spills cost more issue slots than the registers do;
//   * the traversal is the uniform op stream of tree_program.hpp (fused form: the two tips of a cherry on equally long
//     branches are one CHERRY op sharing their exponentials) -- no lane divergence inside a likelihood evaluation;
//     lanes differ only in how many Newton steps they need, and a lane whose column has converged immediately takes
//     the next column of the wave's work share (lane refill), so the wave keeps 64 live columns until it runs dry;
//   * a column's tip states are 4-bit masks packed 8 per word by classify_kernel: in registers for up to 64 tips,
//     streamed one word ahead beyond; the Newton iterations never touch the alignment again;
//   * the running partial (value, d/du, d2/du2 of the 4 conditional likelihoods; u = log siteRate) is 12 doubles
//     in VGPRs, updated IN PLACE (tied-operand asm, see mul_into): no register copies in the op loop; parked
//     siblings go to an LDS stack laid out [slot][component][lane] so every ds_read/write_b64 is a conflict-free
//     512-B row;
//   * tips: the per-mask vectors U^-1 * tip(mask) are a 16x4 table in LDS built once per locus segment;
//   * constant / flat columns never reach this kernel: classify_kernel answers them in closed form and
//     compact_kernel packs the remaining columns so that lanes are not idle beside them;
//   * MFMA is not used: every lane's transition matrix is its own; the shared eigenvector products would need a
//     lane = (column, state) layout, and on gfx950 FP64 MFMA shares the FP64 VALU's rate and issue slot (measured).
//     The kernel is FP64-VALU bound; its HBM traffic is ntaxa/2 + 29 bytes per optimised column.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include "tphip.h"
#include "fast_exp.hpp"
#include "gtr_model.hpp"
#include "tree_program.hpp"
#include "site_rate_params.hpp"

namespace tphip {

struct Partial {  // value and first/second derivative (wrt u) of the 4 conditional likelihoods
    double v[4], d1[4], d2[4];
};

// In-place FP64 updates through one-instruction asm statements whose "+v" operand ties result and accumulator to the
// same register.  Written in plain C++ the accumulator's new values land in fresh registers (v_fmac writes its
// addend, which is a temporary) and every arm of the op switch ends in 8-14 v_mov_b64 copies back into the registers
// the loop carries (7 % of the VALU instructions of an evaluation).  FP64 VALU -> VALU dependencies are
// interlocked in hardware, so the statements need no padding (cdna_hip_programming.md section 5.7 item 2).
__device__ __forceinline__ void mul_into(double& acc, double b) {          // acc *= b
    asm("v_mul_f64 %0, %0, %1" : "+v"(acc) : "v"(b));
}
__device__ __forceinline__ void fma_into(double& acc, double a, double b) {  // acc += a * b
    asm("v_fmac_f64 %0, %1, %2" : "+v"(acc) : "v"(a), "v"(b));
}
// dst = a * b + c where dst's previous value is dead: the "+v" only pins the result to dst's register.
__device__ __forceinline__ void fma_over(double& dst, double a, double b, double c) {
    asm("v_fma_f64 %0, %1, %2, %3" : "+v"(dst) : "v"(a), "v"(b), "v"(c));
}
__device__ __forceinline__ void mul_over(double& dst, double a, double b) {
    asm("v_mul_f64 %0, %1, %2" : "+v"(dst) : "v"(a), "v"(b));
}

// acc *= m (product rule for value / first / second derivative), every component updated in its own register
__device__ __forceinline__ void partial_mul(Partial& a, const Partial& m) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const double twice = a.d1[i] + a.d1[i];
        mul_into(a.d2[i], m.v[i]);
        fma_into(a.d2[i], a.v[i], m.d2[i]);
        fma_into(a.d2[i], twice, m.d1[i]);
        mul_into(a.d1[i], m.v[i]);
        fma_into(a.d1[i], a.v[i], m.d1[i]);
        mul_into(a.v[i], m.v[i]);
    }
}

// The locus' model as plain scalars (wave-uniform: the compiler keeps them in SGPRs).
struct ModelRegs {
    double lam[3], U[12], Ui[12], pi[4];
    double c4;   // 1/24 for exp_nonpos_tab, pinned to a VGPR (see fast_exp.hpp)
};

// The 31 doubles go through LDS so that they land in VGPRs (replicated across lanes): 62 SGPRs of model
// plus the op pipeline and pointers do not fit the scalar file, and the resulting SGPR spills (v_readlane /
// s_mov traffic in every op) cost more issue slots than 62 VGPRs do at the 2 waves/SIMD the LDS stack allows.
__device__ __forceinline__ ModelRegs load_model(const LocusModel* __restrict__ M, double* mtab, int lane) {
    if (lane < 32) mtab[lane] = reinterpret_cast<const double*>(M)[lane];  // lam[3] U[12] Ui[12] pi[4] kappa
    __syncthreads();
    ModelRegs R;
#pragma unroll
    for (int i = 0; i < 3; ++i) R.lam[i] = mtab[i];
#pragma unroll
    for (int i = 0; i < 12; ++i) { R.U[i] = mtab[3 + i]; R.Ui[i] = mtab[15 + i]; }
#pragma unroll
    for (int i = 0; i < 4; ++i) R.pi[i] = mtab[27 + i];
    R.c4 = 4.1666666666666664e-02;
    asm volatile("" : "+v"(R.c4));   // opaque: the compiler keeps it in its register instead of re-creating the constant
    return R;
}

// Mixed-loci mode: every lane carries the model of ITS column's locus, read straight from the (L2-resident) model rows.
// Called under divergence: only the lanes whose locus changes load.
__device__ __forceinline__ void load_model_lane(const LocusModel* __restrict__ M, ModelRegs& R) {
    const double* __restrict__ m = reinterpret_cast<const double*>(M);   // lam[3] U[12] Ui[12] pi[4] kappa
#pragma unroll
    for (int i = 0; i < 3; ++i) R.lam[i] = m[i];
#pragma unroll
    for (int i = 0; i < 12; ++i) { R.U[i] = m[3 + i]; R.Ui[i] = m[15 + i]; }
#pragma unroll
    for (int i = 0; i < 4; ++i) R.pi[i] = m[27 + i];
}

// message of a tip through its branch: P(t s) * tip, with derivatives wrt u = log s.
// w[k] = (U^-1 tip)_k from the LDS mask table; x_k = lam_k t s; e_k = exp(x_k).
__device__ __forceinline__ void tip_message(const ModelRegs& R, const double* __restrict__ etab, const double* w, double ts, Partial& m) {
    double a[3], b[3], c[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        double x = R.lam[k] * ts;
        double e = kUseExpTable ? exp_nonpos_tab(x, etab, R.c4) : exp_nonpos(x);
        a[k] = e * w[k + 1];
        b[k] = x * a[k];
        c[k] = fma(x, b[k], b[k]);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const double* Ur = R.U + i * 3;
        m.v[i] = fma(Ur[2], a[2], fma(Ur[1], a[1], fma(Ur[0], a[0], w[0])));
        m.d1[i] = fma(Ur[2], b[2], fma(Ur[1], b[1], Ur[0] * b[0]));
        m.d2[i] = fma(Ur[2], c[2], fma(Ur[1], c[1], Ur[0] * c[0]));
    }
}

// U * (a, b, c) written over `acc` (whose previous content is dead), results pinned to the registers the loop carries
__device__ __forceinline__ void message_over(const ModelRegs& R, const double (&a)[3], const double (&b)[3], const double (&c)[3],
                                             double w0, Partial& acc) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const double* Ur = R.U + i * 3;
        fma_over(acc.v[i], Ur[0], a[0], w0);
        fma_into(acc.v[i], Ur[1], a[1]);
        fma_into(acc.v[i], Ur[2], a[2]);
        mul_over(acc.d1[i], Ur[0], b[0]);
        fma_into(acc.d1[i], Ur[1], b[1]);
        fma_into(acc.d1[i], Ur[2], b[2]);
        mul_over(acc.d2[i], Ur[0], c[0]);
        fma_into(acc.d2[i], Ur[1], c[1]);
        fma_into(acc.d2[i], Ur[2], c[2]);
    }
}

// TIP_SET: the tip's message becomes the accumulator
__device__ __forceinline__ void tip_message_over(const ModelRegs& R, const double* __restrict__ etab, const double* w, double ts, Partial& acc) {
    double a[3], b[3], c[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        double x = R.lam[k] * ts;
        double e = kUseExpTable ? exp_nonpos_tab(x, etab, R.c4) : exp_nonpos(x);
        a[k] = e * w[k + 1];
        b[k] = x * a[k];
        c[k] = fma(x, b[k], b[k]);
    }
    message_over(R, a, b, c, w[0], acc);
}

// CHERRY: two tips on equally long branches start a subtree; exp(lambda_k t s) is computed once for both messages
// (w1 -> accumulator, w2 -> multiplied in).
__device__ __forceinline__ void cherry_over(const ModelRegs& R, const double* __restrict__ etab, const double* w1, const double* w2,
                                            double ts, Partial& acc) {
    double x[3], e[3], a[3], b[3], c[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        x[k] = R.lam[k] * ts;
        e[k] = kUseExpTable ? exp_nonpos_tab(x[k], etab, R.c4) : exp_nonpos(x[k]);
        a[k] = e[k] * w1[k + 1];
        b[k] = x[k] * a[k];
        c[k] = fma(x[k], b[k], b[k]);
    }
    message_over(R, a, b, c, w1[0], acc);
    Partial m;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        a[k] = e[k] * w2[k + 1];
        b[k] = x[k] * a[k];
        c[k] = fma(x[k], b[k], b[k]);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const double* Ur = R.U + i * 3;
        m.v[i] = fma(Ur[2], a[2], fma(Ur[1], a[1], fma(Ur[0], a[0], w2[0])));
        m.d1[i] = fma(Ur[2], b[2], fma(Ur[1], b[1], Ur[0] * b[0]));
        m.d2[i] = fma(Ur[2], c[2], fma(Ur[1], c[1], Ur[0] * c[0]));
    }
    partial_mul(acc, m);
}

// acc <- P(t s) * acc with derivatives (internal branch)
__device__ __forceinline__ void branch_apply(const ModelRegs& R, const double* __restrict__ etab, double ts, Partial& p) {
    double w0[4], w1[4], w2[4];
    w0[0] = fma(R.pi[3], p.v[3], fma(R.pi[2], p.v[2], fma(R.pi[1], p.v[1], R.pi[0] * p.v[0])));
    w1[0] = fma(R.pi[3], p.d1[3], fma(R.pi[2], p.d1[2], fma(R.pi[1], p.d1[1], R.pi[0] * p.d1[0])));
    w2[0] = fma(R.pi[3], p.d2[3], fma(R.pi[2], p.d2[2], fma(R.pi[1], p.d2[1], R.pi[0] * p.d2[0])));
#pragma unroll
    for (int k = 1; k < 4; ++k) {
        const double* Ir = R.Ui + (k - 1) * 4;
        w0[k] = fma(Ir[3], p.v[3], fma(Ir[2], p.v[2], fma(Ir[1], p.v[1], Ir[0] * p.v[0])));
        w1[k] = fma(Ir[3], p.d1[3], fma(Ir[2], p.d1[2], fma(Ir[1], p.d1[1], Ir[0] * p.d1[0])));
        w2[k] = fma(Ir[3], p.d2[3], fma(Ir[2], p.d2[2], fma(Ir[1], p.d2[1], Ir[0] * p.d2[0])));
    }
    const double z0 = w0[0], z1 = w1[0], z2 = w2[0];  // eigenvalue 0: e = 1
    double a[3], b[3], c[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        double x = R.lam[k] * ts;
        double e = kUseExpTable ? exp_nonpos_tab(x, etab, R.c4) : exp_nonpos(x);
        a[k] = e * w0[k + 1];
        double ew1 = e * w1[k + 1], ew2 = e * w2[k + 1];
        b[k] = fma(x, a[k], ew1);
        c[k] = fma(x, b[k] + a[k] + ew1, ew2);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const double* Ur = R.U + i * 3;
        fma_over(p.v[i], Ur[0], a[0], z0);   // p's previous content was consumed by the w's above
        fma_into(p.v[i], Ur[1], a[1]);
        fma_into(p.v[i], Ur[2], a[2]);
        fma_over(p.d1[i], Ur[0], b[0], z1);
        fma_into(p.d1[i], Ur[1], b[1]);
        fma_into(p.d1[i], Ur[2], b[2]);
        fma_over(p.d2[i], Ur[0], c[0], z2);
        fma_into(p.d2[i], Ur[1], c[1]);
        fma_into(p.d2[i], Ur[2], c[2]);
    }
}

// Rescaling for deep trees / hundreds of taxa: when the largest component of the running partial is below 2^-256
// the partial (value and both derivatives) is multiplied by 2^-e and e is added to `scale` -- exact, and everything
// downstream is linear in it.  The test runs before every TIP_MUL and BRANCH; it is 4 integer instructions and a
// ballot (partials are non-negative, so their high words order like unsigned integers), and the multiplication
// sits behind a wave-uniform branch that is almost never taken: 12 in-place multiplies there instead of 4-6
// multiplies and 6 more integer instructions folded into every message.
__device__ __forceinline__ void rescale_if_needed(Partial& p, int& scale) {
    const unsigned h0 = (unsigned)__double2hiint(p.v[0]), h1 = (unsigned)__double2hiint(p.v[1]);
    const unsigned h2 = (unsigned)__double2hiint(p.v[2]), h3 = (unsigned)__double2hiint(p.v[3]);
    const unsigned mx = max(max(h0, h1), max(h2, h3));
    const unsigned kOne = 1u << 20;                      // smallest normal high word
    const unsigned kLow = (unsigned)(1023 - 256) << 20;  // 2^-256
    const bool need = (mx - kOne) < (kLow - kOne);       // below 2^-256 and still a normal number
    if (__any(need)) {
        const int e = need ? (int)(mx >> 20) - 1023 : 0;
        scale += e;
        const double f = __hiloint2double((1023 - e) << 20, 0);   // 2^-e
#pragma unroll
        for (int i = 0; i < 4; ++i) { mul_into(p.v[i], f); mul_into(p.d1[i], f); mul_into(p.d2[i], f); }
    }
}

// Hand-issued scalar load of one TreeOp (16 bytes) and its explicit wait; see evaluate_column.
typedef int SOp __attribute__((ext_vector_type(4)));
__device__ __forceinline__ SOp sop_issue(const TreeOp* p) {  // p must be wave-uniform
    SOp r;
    asm volatile("s_load_dwordx4 %0, %1, 0x0" : "=s"(r) : "s"(p));
    return r;
}
__device__ __forceinline__ void sop_wait(SOp& r) {  // ties every later use of r behind the wait
    asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(r));
}
__device__ __forceinline__ TreeOp sop_decode(const SOp& r) {
    TreeOp o;
    o.code = r.x;
    o.taxon = r.y;
    o.t = __hiloint2double(r.w, r.z);
    return o;
}

__device__ __forceinline__ unsigned load_state(const SiteParams& P, int taxon, int64_t col) {
    return P.states[(int64_t)taxon * P.ncols_total + col];
}

// Word `w` (wave-uniform index) of the lane's packed tip states.  The index is scalar, so this is a chain of
// scalar compares feeding v_cndmask -- executed once per 8 tips.
// `pk[i]` for a compile-time i, through an opaque copy: keeps the words in VGPRs (without it the selection below
// is rewritten as a dynamically indexed scratch array).
template <int NW>
__device__ __forceinline__ uint32_t word_at(const uint32_t (&pk)[NW > 0 ? NW : 1], int i) {
    uint32_t v = pk[i];
    asm volatile("" : "+v"(v));
    return v;
}

// Word `w` (wave-uniform index) of the lane's packed tip states; executed once per 8 tips.  Up to 8 words: a
// chain of scalar compares feeding v_cndmask.  More words: scalar branches on the high bits select a group
// of 8 first, so the cost stays ~8 selects instead of NW.
template <int NW>
__device__ __forceinline__ uint32_t pick_word(const uint32_t (&pk)[NW > 0 ? NW : 1], int w) {
    if constexpr (NW <= 8) {
        uint32_t r = word_at<NW>(pk, 0);
#pragma unroll
        for (int i = 1; i < NW; ++i) { const uint32_t c = word_at<NW>(pk, i); r = (w == i) ? c : r; }
        return r;
    } else {
        const int grp = __builtin_amdgcn_readfirstlane(w >> 3), sub = w & 7;
        uint32_t r = 0;
#pragma unroll
        for (int g = 0; g < NW / 8; ++g) {
            if (grp == g) {  // wave-uniform: a scalar branch
                r = word_at<NW>(pk, 8 * g);
#pragma unroll
                for (int i = 1; i < 8; ++i) { const uint32_t c = word_at<NW>(pk, 8 * g + i); r = (sub == i) ? c : r; }
            }
        }
        return r;
    }
}

// One likelihood evaluation for this lane's column: f = log L, g = df/du, h = d2f/du2 at s = exp(u).
// NW > 0: the column's tip states live in registers (pk: 8 four-bit masks per word, in program tip order), so
//         the Newton iterations never touch the alignment again;
// NW = kStreamWords: more than 64 tips -- the packed words would take 16+ registers and push the kernel past 256
//         VGPRs (one wave per SIMD).  Only word 0 stays in a register; word w + 1 is requested from the packed array
//         (L2-resident: 4 bytes per 8 tips per evaluation) when word w is started, i.e. eight ops before it is needed;
// NW = 0: states are fetched from the byte array one op ahead (the eval_columns diagnostic, which thereby
//         cross-checks the packed paths, and TPHIP_FORCE_BYTE_PATH).
// SPILL: the parked partials beyond the first P.lds_depth live in a global scratch row of this wave (L2-resident)
//        instead of LDS.  On a deep tree (256 taxa: 4 parked partials = 24 KB of LDS per wave) the LDS stack, not the
//        registers, caps the CU at 6 waves; the deepest slot is used by ~1 push in 9, so parking it in global memory
//        costs little and lets 8 waves (2 per SIMD) stay resident.
template <int NW, bool SPILL = false>
__device__ __forceinline__ void evaluate_column(const SiteParams& P, const ModelRegs& R, const double* __restrict__ wtab,
                                                const double* __restrict__ etab, double* __restrict__ stack, int64_t col,
                                                const uint32_t (&pk)[NW > 0 ? NW : 1], double s, double& f,
                                                double& g, double& h) {
    Partial acc;
    int scale = 0;
    int sp = 0;
    int tk = 0;          // tips consumed so far (wave-uniform)
    uint32_t cur = 0;    // current packed word, already shifted
    const int lane = threadIdx.x;
#pragma unroll
    for (int i = 0; i < 4; ++i) { acc.v[i] = 1.0; acc.d1[i] = 0.0; acc.d2[i] = 0.0; }
    // The op stream is fetched two ops ahead by hand-issued scalar loads (s_load_dwordx4 = one 16-byte TreeOp).
    // Through plain C++ the compiler either turns the fetch into a vector load (the kernel also stores, so the
    // pointer is not provably invariant) or, via the constant address space, into an s_load that it waits for
    // immediately; issued here and waited for at the END of the op (sop_wait), the ~100-cycle scalar-cache round
    // trip disappears under the op's FP64 work.  A hidden SMEM op in flight only makes the compiler's own
    // lgkmcnt waits for LDS more conservative, never unsafe (cdna_hip_programming.md section 5.7).
    const TreeOp* __restrict__ ops = P.ops;
    const int last = P.nops - 1;
    TreeOp op, nxt;
    {
        SOp r0 = sop_issue(ops), r1 = sop_issue(ops + (last < 1 ? last : 1));
        sop_wait(r0);
        sop_wait(r1);
        op = sop_decode(r0);
        nxt = sop_decode(r1);
    }
    uint32_t nxtw = 0;   // streamed path: the word after the current one, in flight
    if constexpr (NW == kStreamWords) nxtw = pk[0];
    auto next_mask = [&]() -> unsigned {   // the next tip's 4-bit mask, in program order (tk is wave-uniform)
        if constexpr (NW > 0) {
            if ((tk & 7) == 0) cur = pick_word<NW>(pk, tk >> 3);
        } else if constexpr (NW == kStreamWords) {
            if ((tk & 7) == 0) {
                // The request for word w + 1 must stay in flight for the next eight ops: as a plain load the compiler
                // waits for it at the end of this block (the value has to reach the register the loop carries).  Issued
                // and awaited by hand, both statements tied ("+v") to that one register (section 5.7, form ii); word 0
                // never passes through here in flight: pk[0] is an ordinary load.
                if (tk > 0) asm volatile("s_waitcnt vmcnt(0)" : "+v"(nxtw));
                cur = nxtw;
                int w = (tk >> 3) + 1;
                w = w < P.nwords ? w : P.nwords - 1;
                const uint32_t* src = P.packed + ((int64_t)w * P.ncols_total + col);
                asm volatile("global_load_dword %0, %1, off" : "+v"(nxtw) : "v"(src));
            }
        }
        const unsigned m = cur & 15u;
        cur >>= 4;
        ++tk;
        return m;
    };
    unsigned st = 15u;
    if constexpr (NW == 0) st = load_state(P, op.taxon, col);  // the program always starts at a tip
    for (int ip = 0; ip < P.nops; ++ip) {
        // Two-deep fetch pipeline: op ip+2 (scalar load) and, on the byte path, the state of op ip+1 are issued
        // here and consumed one iteration later, i.e. they stay in flight under this op's FP64 work.
        SOp raw = sop_issue(ops + ((ip + 2 < last) ? ip + 2 : last));
        unsigned st_nxt = 15u;
        if constexpr (NW == 0) {
            if ((nxt.code & OP_CODE_MASK) <= OP_TIP_MUL) st_nxt = load_state(P, nxt.taxon, col);
        }
        const int code = op.code & OP_CODE_MASK;   // the fused stream ORs OP_PUSH_BEFORE / OP_POP_AFTER into the code
        auto park = [&]() {     // PUSH: the finished sibling goes to the LDS stack; the TIP_SET / CHERRY that always
            if (SPILL && sp >= P.lds_depth) {   // follows overwrites the accumulator        (sp is wave-uniform)
                double* slot = P.spill + ((size_t)blockIdx.x * (size_t)(P.stack_depth - P.lds_depth) + (size_t)(sp - P.lds_depth)) * 12 * kSiteBlock + lane;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    slot[(i)*kSiteBlock] = acc.v[i];
                    slot[(4 + i) * kSiteBlock] = acc.d1[i];
                    slot[(8 + i) * kSiteBlock] = acc.d2[i];
                }
            } else {
                double* slot = stack + (size_t)sp * 12 * kSiteBlock + lane;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    slot[(i)*kSiteBlock] = acc.v[i];
                    slot[(4 + i) * kSiteBlock] = acc.d1[i];
                    slot[(8 + i) * kSiteBlock] = acc.d2[i];
                }
            }
            ++sp;
        };
        auto pop_mul = [&]() {  // POP_MUL: acc *= parked sibling
            Partial m;
            --sp;
            if (SPILL && sp >= P.lds_depth) {
                const double* slot = P.spill + ((size_t)blockIdx.x * (size_t)(P.stack_depth - P.lds_depth) + (size_t)(sp - P.lds_depth)) * 12 * kSiteBlock + lane;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    m.v[i] = slot[(i)*kSiteBlock];
                    m.d1[i] = slot[(4 + i) * kSiteBlock];
                    m.d2[i] = slot[(8 + i) * kSiteBlock];
                }
            } else {
                const double* slot = stack + (size_t)sp * 12 * kSiteBlock + lane;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    m.v[i] = slot[(i)*kSiteBlock];
                    m.d1[i] = slot[(4 + i) * kSiteBlock];
                    m.d2[i] = slot[(8 + i) * kSiteBlock];
                }
            }
            partial_mul(acc, m);
        };
        if (code == OP_BRANCH) {
            rescale_if_needed(acc, scale);
            branch_apply(R, etab, op.t * s, acc);
            if (op.code & OP_POP_AFTER) pop_mul();
        } else if (code <= OP_TIP_MUL) {
            unsigned mask;
            if constexpr (NW != 0) {
                mask = next_mask();   // zero codes were turned into 15 when the word was packed
            } else {
                mask = st & 15u;
                mask = mask ? mask : 15u;
            }
            const double* w = wtab + mask * 4;
            if (code == OP_TIP_SET) {
                // A subtree starts here (program start, or right after a PUSH: tree_program.hpp emits no other
                // TIP_SET): the tip's message IS the new accumulator -- no product with an identity, no rescaling.
                if (op.code & OP_PUSH_BEFORE) park();
                const double wv[4] = {w[0], w[1], w[2], w[3]};
                tip_message_over(R, etab, wv, op.t * s, acc);
            } else {
                Partial m;
                rescale_if_needed(acc, scale);
                const double wv[4] = {w[0], w[1], w[2], w[3]};
                tip_message(R, etab, wv, op.t * s, m);
                partial_mul(acc, m);
            }
        } else if (NW != 0 && code == OP_CHERRY) {   // fused stream (packed paths only)
            if (op.code & OP_PUSH_BEFORE) park();
            const unsigned mask1 = next_mask();
            const unsigned mask2 = next_mask();
            const double* w1 = wtab + mask1 * 4;
            const double* w2 = wtab + mask2 * 4;
            const double wa[4] = {w1[0], w1[1], w1[2], w1[3]};
            const double wb[4] = {w2[0], w2[1], w2[2], w2[3]};
            cherry_over(R, etab, wa, wb, op.t * s, acc);
        } else if (code == OP_POP_MUL) {
            pop_mul();
        } else {  // OP_PUSH
            park();
        }
        sop_wait(raw);
        op = nxt;
        nxt = sop_decode(raw);
        st = st_nxt;
    }
    // the last word's (clamped, unused) request must have landed before its register is reused
    if constexpr (NW == kStreamWords) asm volatile("s_waitcnt vmcnt(0)" : "+v"(nxtw));
    double L = 0, L1 = 0, L2 = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        L = fma(R.pi[i], acc.v[i], L);
        L1 = fma(R.pi[i], acc.d1[i], L1);
        L2 = fma(R.pi[i], acc.d2[i], L2);
    }
    double inv = 1.0 / L;
    g = L1 * inv;
    h = L2 * inv - g * g;
    f = log(L) + (double)scale * 0.6931471805599453;
}

// The objective the optimiser sees.  ncat <= 1 (the reference's model): one evaluation at s.  ncat > 1: the mixture
// L(s) = sum_k w_k L(s rho_k) -- the "+G" of GTR+G as an opt-in extension the reference's script does not have
// (SURVEY F2).  With f_k, g_k, h_k the category's log-likelihood and u-derivatives and p_k its posterior weight:
// f = logsumexp(log w_k + f_k), g = sum p_k g_k, h = sum p_k (h_k + g_k^2) - g^2 (running maximum, one pass).
template <int NW, bool SPILL = false>
__device__ __forceinline__ void evaluate_site(const SiteParams& P, const ModelRegs& R, const double* __restrict__ wtab,
                                              const double* __restrict__ etab, double* __restrict__ stack, int64_t col,
                                              const uint32_t (&pk)[NW > 0 ? NW : 1], double s, double& f, double& g,
                                              double& h) {
    const int K = P.ncat > 1 ? P.ncat : 1;
    double top = -INFINITY, z = 0.0, a = 0.0, b = 0.0;
    for (int k = 0; k < K; ++k) {
        double fk, gk, hk;
        evaluate_column<NW, SPILL>(P, R, wtab, etab, stack, col, pk, P.ncat > 1 ? s * P.cat[k] : s, fk, gk, hk);
        if (P.ncat <= 1) { f = fk; g = gk; h = hk; return; }
        fk += P.cat[K + k];
        if (fk > top) {
            const double r = exp(top - fk);   // exp(-inf) = 0 on the first category
            z *= r; a *= r; b *= r;
            top = fk;
        }
        const double p = exp(fk - top);
        z += p;
        a = fma(p, gk, a);
        b = fma(p, fma(gk, gk, hk), b);
    }
    f = top + log(z);
    g = a / z;
    h = b / z - g * g;
}

// tip table: wtab[mask][k] = sum_{j in mask} U^-1[k][j]   (row 0 of U^-1 is pi)
__device__ __forceinline__ void build_tip_table(const LocusModel* __restrict__ M, double* wtab, int lane) {
    int mask = lane >> 2, k = lane & 3;
    double w = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        double uij = (k == 0) ? M->pi[j] : M->Ui[(k - 1) * 4 + j];
        w += ((mask >> j) & 1) ? uij : 0.0;
    }
    wtab[lane] = w;
}

// Exclusive scans of the per-locus work counts and of the per-locus slice counts (one workgroup; L is at most a
// few 10^4).
__global__ __launch_bounds__(1024) void scan_counts_kernel(const int32_t* __restrict__ count, int64_t nloci, int32_t chunk_cols,
                                                           int64_t* __restrict__ prefix, int64_t* __restrict__ slice_prefix) {
    __shared__ long long wave_sum[16], wave_ssum[16];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int64_t per = (nloci + 1023) / 1024;
    const int64_t lo = t * per, hi = (lo + per < nloci) ? lo + per : nloci;
    long long s = 0, q = 0;
    for (int64_t i = lo; i < hi; ++i) { s += count[i]; q += site_slices(count[i], chunk_cols); }
    // block-wide exclusive scan of the 1024 per-thread sums: shuffle scan inside each wave, then over the 16 wave
    // totals (a single thread walking 1024 LDS entries took 20 us: 4 % of a C2 step)
    long long inc = s, sinc = q;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const long long a = __shfl_up(inc, o), b = __shfl_up(sinc, o);
        if (lane >= o) { inc += a; sinc += b; }
    }
    if (lane == 63) { wave_sum[wave] = inc; wave_ssum[wave] = sinc; }
    __syncthreads();
    long long base = 0, sbase = 0, total = 0, stotal = 0;
#pragma unroll
    for (int w = 0; w < 16; ++w) {
        const long long a = wave_sum[w], b = wave_ssum[w];
        if (w < wave) { base += a; sbase += b; }
        total += a; stotal += b;
    }
    if (t == 0) { prefix[nloci] = total; slice_prefix[nloci] = stotal; }
    long long run = base + inc - s, srun = sbase + sinc - q;
    for (int64_t i = lo; i < hi; ++i) {
        prefix[i] = run; run += count[i];
        slice_prefix[i] = srun; srun += site_slices(count[i], chunk_cols);
    }
}

// Persistent, statically balanced launch: the grid is exactly the number of waves the chip keeps resident;
// wave w takes the w-th EQUAL share of the batch's global work list (the concatenation of every locus'
// compacted column list, located through the prefix sums of the per-locus counts).  Inside one locus segment
// the wave refills freed lanes continuously; it drains only at a locus boundary (the model changes) and at the
// end of its share.  Equal column counts balance the waves to ~2 % (thousands of columns per wave), so there
// is neither a partially filled last round of workgroups nor a tail of short slices.  No wave waits on
// another: every wave leaves after its own share, whatever the others do.
#ifndef TPHIP_SITE_MIN_WAVES
#define TPHIP_SITE_MIN_WAVES 1
#endif
//
// MIXED (small batches of short loci, e.g. 1000 loci x 500 columns): the persistent scheme with EQUAL shares, but a wave
// carries the columns of up to kMixedLoci loci at once -- every lane holds the model of its own column's locus (the model
// registers are lane-replicated VGPRs anyway, so this costs no register) and reads the tip table of that locus (one 512-byte
// table per locus of the group in LDS).  A wave per locus-aligned slice, the other small-batch mode, leaves the chip to the
// slowest locus (27 rounds of evaluations against 18 on average on C2) with one wave per SIMD and a drain per locus; here
// shares are equal column counts whatever the loci, lanes are refilled across locus boundaries and a wave drains once.
// Results are bit-identical: a column's arithmetic does not depend on the lane or wave that carries it.
template <int NW, bool SPILL = false, bool MIXED = false>
__global__ __launch_bounds__(kSiteBlock, TPHIP_SITE_MIN_WAVES) void site_rate_kernel(SiteParams P) {
    extern __shared__ double lds[];
    double* wtab = lds;          // [16 masks][4]          MIXED: [kMixedLoci][16][4]
    double* mtab = lds + 64;     // [32] the locus' model  MIXED: unused, the segment table sits behind etab
    double* etab = lds + (MIXED ? kMixedLoci * 64 : 96);     // [64] 2^(j/64)
    double* stack = lds + (MIXED ? kMixedLdsHeader : kSiteLdsHeader);  // [stack_depth][12][64]
    // MIXED: the loci of the current group: where a group-relative work index finds its column (work[seg_off + idx]), the
    // index at which the locus' part ends, the locus
    int64_t* seg_off = reinterpret_cast<int64_t*>(lds + kMixedLoci * 64 + 64);
    int* seg_end = reinterpret_cast<int*>(seg_off + kMixedLoci);
    int* seg_locus = seg_end + kMixedLoci;
    int* seg_nh = seg_locus + kMixedLoci;     // slow columns at the head of the locus' part of the reordered list
    int* seg_hcum = seg_nh + kMixedLoci;      // ... summed over the parts up to and including this one
    int* seg_ecum = seg_hcum + kMixedLoci;    // the same for the other columns
    const int lane = threadIdx.x;
    etab[lane] = kExp2Table[lane];
    int64_t g0, g1, lo_l;
    if (P.persistent) {
        const int64_t total = P.work_prefix[P.nloci];
        // MIXED: one wave per SIMD or two is decided here, where the length of the work list is known (after classification
        // and de-duplication): a short list ends with the wave that carries its slowest column, and a second wave on that
        // wave's SIMD makes each of its rounds longer (tphip.hip, where the threshold is set)
        unsigned nshares = gridDim.x;
        if constexpr (MIXED) {
            if (total < P.mixed_switch_cols && nshares > (unsigned)P.mixed_few_waves) nshares = (unsigned)P.mixed_few_waves;
            if (blockIdx.x >= nshares) return;
        }
        // Share boundaries: with more shares than resident waves the first round (one share per resident wave) gets
        // `first_fraction` of the work and the later, smaller shares are what the dispatcher balances the waves'
        // finishing times with (equal shares when first_fraction = first_round / gridDim.x).
        auto boundary = [&](unsigned b) -> int64_t {
            const unsigned R = (unsigned)P.first_round, N = nshares;
            if (R >= N) return (int64_t)b * total / N;
            const double f = (b <= R) ? P.first_fraction * (double)b / (double)R
                                      : P.first_fraction + (1.0 - P.first_fraction) * (double)(b - R) / (double)(N - R);
            const int64_t g = (int64_t)(f * (double)total);
            return b >= N ? total : (g < total ? g : total);
        };
        g0 = boundary(blockIdx.x);
        g1 = boundary(blockIdx.x + 1);
        if (g0 >= g1) return;
        // first locus of this share: the last l with prefix[l] <= g0 (binary search, wave-uniform)
        int64_t hi_l = P.nloci;
        lo_l = 0;
        while (hi_l - lo_l > 1) {
            const int64_t mid = (lo_l + hi_l) >> 1;
            if (P.work_prefix[mid] <= g0) lo_l = mid; else hi_l = mid;
        }
    } else {
        // Small batches (a share would be a few hundred columns): one workgroup per locus-aligned slice instead,
        // because cutting a small locus in two doubles its prologue and drain.  A locus' `count` columns are split
        // into site_slices(count) EQUAL slices; workgroup b takes the b-th slice of the batch, located through the
        // prefix sums of the per-locus slice counts.  The grid was sized for the worst case (every column needs the
        // optimiser); the surplus workgroups are the LAST ones of the grid and leave at once -- interleaved with the
        // working ones (one fixed slot per possible slice) they alias with the round-robin of workgroups over the 8
        // XCDs: with two slots per locus and one in use, four XCDs did all the work.
        const int64_t nslices = P.slice_prefix[P.nloci];
        if ((int64_t)blockIdx.x >= nslices) return;
        int64_t hi_l = P.nloci;
        lo_l = 0;
        while (hi_l - lo_l > 1) {
            const int64_t mid = (lo_l + hi_l) >> 1;
            if (P.slice_prefix[mid] <= (int64_t)blockIdx.x) lo_l = mid; else hi_l = mid;
        }
        const int count = P.work_count[lo_l];
        const int ns = site_slices(count, P.chunk_cols);
        const int j = (int)((int64_t)blockIdx.x - P.slice_prefix[lo_l]);
        const int64_t pbeg = P.work_prefix[lo_l];
        g0 = pbeg + (int64_t)j * count / ns;
        g1 = pbeg + (int64_t)(j + 1) * count / ns;
        if (P.work_cols2) {
            // Slow columns first (classify_kernel marked them in the lnl slot): a stable partition of this wave's slice of
            // the work list into the scratch list, two passes of 64 entries at a time.  A wave lasts until its slowest lane
            // is done; with ~5 columns per lane, a 15-evaluation column handed out in the last round doubles that.
            const int32_t* __restrict__ src = P.work_cols + P.locus_offsets[lo_l];
            int32_t* __restrict__ dst = P.work_cols2 + P.locus_offsets[lo_l];
            const int b = (int)(g0 - pbeg), e = (int)(g1 - pbeg);
            int nh = 0;
            for (int i = b; i < e; i += kSiteBlock) {
                const bool in = i + lane < e;
                const int32_t c = src[in ? i + lane : b];
                nh += __popcll(__ballot(in && P.lnl[c] != 0.0));
            }
            int ph = b, pe = b + nh;
            for (int i = b; i < e; i += kSiteBlock) {
                const bool in = i + lane < e;
                const int32_t c = src[in ? i + lane : b];
                const bool hard = in && P.lnl[c] != 0.0;
                const unsigned long long mh = __ballot(hard), me = __ballot(in && !hard);
                const unsigned long long below = (1ull << lane) - 1ull;
                if (hard) dst[ph + __popcll(mh & below)] = c;
                else if (in) dst[pe + __popcll(me & below)] = c;
                ph += __popcll(mh);
                pe += __popcll(me);
            }
            __threadfence_block();
            __syncthreads();
        }
    }
    unsigned evals = 0;
#ifdef TPHIP_SITE_TRACE_ROUNDS
    unsigned rounds = 0;
    const unsigned long long tick0 = wall_clock64();
#endif
    int64_t gpos = g0;
    int64_t locus = lo_l;
    while (gpos < g1) {
        ModelRegs R;
        const double* wt = wtab;     // the lane's tip table
        double kappa = 0.0;
        int begin, end;
        int myk = 0, K = 1;          // MIXED: the lane's locus within the group; loci in the group
        const int32_t* __restrict__ work;
        if constexpr (!MIXED) {
            const int64_t pbeg = P.work_prefix[locus], pend = P.work_prefix[locus + 1];
            const int64_t seg_end_g = pend < g1 ? pend : g1;
            if (seg_end_g <= gpos) { ++locus; continue; }  // locus without optimiser work
            begin = (int)(gpos - pbeg);
            end = (int)(seg_end_g - pbeg);
            gpos = seg_end_g;
            __syncthreads();  // the previous segment's readers of wtab / mtab are done
            const LocusModel* __restrict__ M = P.models + locus;
            build_tip_table(M, wtab, lane);
            R = load_model(M, mtab, lane);
            kappa = mtab[31];
            work = ((!P.persistent && P.work_cols2) ? (const int32_t*)P.work_cols2 : P.work_cols) + P.locus_offsets[locus];
            ++locus;
        } else {
            __syncthreads();  // the previous group's readers of the tables are done
            const int64_t gb = gpos;
            K = 0;
            while (K < kMixedLoci && gpos < g1) {
                const int64_t pbeg = P.work_prefix[locus], pend = P.work_prefix[locus + 1];
                const int64_t se = pend < g1 ? pend : g1;
                if (se > gpos) {
                    if (lane == 0) {
                        seg_off[K] = P.locus_offsets[locus] - pbeg + gb;
                        seg_end[K] = (int)(se - gb);
                        seg_locus[K] = (int)locus;
                    }
                    build_tip_table(P.models + locus, wtab + 64 * K, lane);
                    ++K;
                    gpos = se;
                }
                if (pend <= gpos) ++locus;
            }
            begin = 0;
            end = (int)(gpos - gb);
            __threadfence_block();
            __syncthreads();
            if (P.work_cols2) {   // slow columns first inside every locus' part (see the small-batch mode above)
                for (int k = 0; k < K; ++k) {
                    const int32_t* __restrict__ src = P.work_cols + seg_off[k];
                    int32_t* __restrict__ dst = P.work_cols2 + seg_off[k];
                    const int b = k ? seg_end[k - 1] : 0, e = seg_end[k];
                    int nh = 0;
                    for (int i = b; i < e; i += kSiteBlock) {
                        const bool in = i + lane < e;
                        const int32_t c = src[in ? i + lane : b];
                        nh += __popcll(__ballot(in && P.lnl[c] != 0.0));
                    }
                    if (lane == 0) seg_nh[k] = nh;
                    int ph = b, pe = b + nh;
                    for (int i = b; i < e; i += kSiteBlock) {
                        const bool in = i + lane < e;
                        const int32_t c = src[in ? i + lane : b];
                        const bool hard = in && P.lnl[c] != 0.0;
                        const unsigned long long mh = __ballot(hard), me = __ballot(in && !hard);
                        const unsigned long long below = (1ull << lane) - 1ull;
                        if (hard) dst[ph + __popcll(mh & below)] = c;
                        else if (in) dst[pe + __popcll(me & below)] = c;
                        ph += __popcll(mh);
                        pe += __popcll(me);
                    }
                }
            } else if (lane < K) {
                seg_nh[lane] = 0;
            }
            __threadfence_block();
            __syncthreads();
            if (lane == 0) {   // the slow columns of ALL parts are handed out first, then the rest, part by part
                int hc = 0, ec = 0;
                for (int k = 0; k < K; ++k) {
                    const int b = k ? seg_end[k - 1] : 0;
                    hc += seg_nh[k];
                    ec += seg_end[k] - b - seg_nh[k];
                    seg_hcum[k] = hc;
                    seg_ecum[k] = ec;
                }
            }
            __threadfence_block();
            __syncthreads();
            work = P.work_cols2 ? (const int32_t*)P.work_cols2 : P.work_cols;
            R.c4 = 4.1666666666666664e-02;
            asm volatile("" : "+v"(R.c4));
        }
        // MIXED: the column at group-relative work index idx, with its locus' position in the group in the top four bits
        auto fetch = [&](int idx) -> int32_t {
            if constexpr (!MIXED) return work[idx];
            const int H = seg_hcum[K - 1];
            const bool slow = idx < H;
            const int e = slow ? idx : idx - H;
            const int* cum = slow ? seg_hcum : seg_ecum;
            int k = 0;
            for (int j = 0; j + 1 < K; ++j) k += (e >= cum[j]) ? 1 : 0;
            // position in the (reordered) list of part k: its slow columns come first
            const int pos = (k ? seg_end[k - 1] : 0) + (slow ? 0 : seg_nh[k]) + e - (k ? cum[k - 1] : 0);
            return work[seg_off[k] + pos] | (k << 28);
        };

        int next = begin + kSiteBlock;  // wave-uniform: first work index not yet handed to a lane
        bool done = (begin + lane >= end);
        int64_t col;
        {
            const int32_t c0 = fetch(done ? begin : begin + lane);
            col = MIXED ? (c0 & 0x0fffffff) : c0;
            if constexpr (MIXED) {
                myk = c0 >> 28;
                load_model_lane(P.models + seg_locus[myk], R);
                wt = wtab + 64 * myk;
            }
        }
        // the next 64 candidates of the work list, one per lane, requested an evaluation before a refill needs them: the
        // refill then costs one memory round trip (the column's packed tips and start value), not two
        int32_t cand = fetch(next + lane < end ? next + lane : end - 1);
        uint32_t pk[NW > 0 ? NW : 1] = {0};
#pragma unroll
        for (int w = 0; w < NW; ++w) pk[w] = (w < P.nwords) ? P.packed[(int64_t)w * P.ncols_total + col] : 0u;
        if constexpr (NW == kStreamWords) pk[0] = P.packed[col];
        // start: classify_kernel left the column's parsimony-based log rate in its `rate` slot (pi_kernels.hpp)
        // ... and its parsimony length m in the `subst` slot (0: not available).  Until a previous point exists the h_prev
        // register carries m: the first step uses it (below) and no register is added to a kernel at 250 VGPRs.
        double u = done ? 0.0 : P.rate[col], lo = kUMin, hi = kUMax, u_prev = 0.0, h_prev = done ? 0.0 : P.subst[col], g_prev = 0.0;
        bool lo_open = true, hi_open = true, have_prev = false;
        bool checking = false;            // this evaluation is the saturation check at kUMax
        double u_conv = 0.0, f_conv = 0.0;
        int it = 0;
        while (true) {
            double f, g, h;
            evaluate_site<NW, SPILL>(P, R, wt, etab, stack, col, pk, exp(u), f, g, h);
#ifdef TPHIP_SITE_TRACE_ROUNDS
            ++rounds;
#endif
            if (!done) {
                const double f_eval = f;
                ++evals;
                ++it;
                int flg = -1;
                const bool uphill = !(g <= 0.0);
                if (checking) {
                    // value at the largest rate vs the converged optimum
                    if (f >= f_conv - kSatTol * fmax(1.0, fabs(f_conv))) { flg = TPHIP_FLAG_SATURATED; u = kUMax; }
                    else { flg = TPHIP_FLAG_OK; u = u_conv; f = f_conv; }
                    checking = false;
                } else
                // Saturation: beyond this point g is second-order small under first-order rounding noise, its
                // sign is meaningless; report the policy value s = 1e4 (same rule as the oracle).
                if (fabs(g) < kFlatEps && fabs(h) < kFlatEps) { flg = TPHIP_FLAG_SATURATED; u = kUMax; }
                else if (u >= kUMax && uphill) { flg = TPHIP_FLAG_SATURATED; }
                else if (u <= kUMin && !uphill) { flg = TPHIP_FLAG_ZERO; }
                else {
                    if (uphill) { lo = u; lo_open = false; } else { hi = u; hi_open = false; }
                    // Two evaluations in hand: f' is known with its slope at both points and its integral between them (the
                    // difference of the values): the quartic through those five conditions locates the zero to fifth order, so
                    // a remaining step below kHermiteTol is taken without evaluating again (same rule as the oracle, which
                    // documents the bounds).  Only between REGULAR points (|h| >= kHermiteRegular): a weakly curved column sits
                    // near an inflection or a plateau, no low-order model of f' holds there, it iterates until the step itself
                    // is below kStepTolFirst.  The previous value lives in f_conv, which has no other use until a saturation
                    // check starts.
                    bool hermite = false;
                    double un = u;
                    const bool regular_here = fabs(h) >= kHermiteRegular;
                    if (have_prev && h < 0.0 && regular_here && fabs(h_prev) >= kHermiteRegular) {
                        const double d = u - u_prev;      // the previous point sits at t = -d
                        const double id = 1.0 / d, id2 = id * id;
                        const double r1 = fma(h, d, g_prev - g), r2 = (h_prev - h) * d, r3 = fma(0.5 * h, d, (f - f_conv) * id - g);
                        const double q2 = fma(30.0, r3, fma(-1.5, r2, -12.0 * r1)) * id2;
                        const double q3 = fma(60.0, r3, fma(-4.0, r2, -28.0 * r1)) * (id2 * id);
                        const double q4 = fma(30.0, r3, fma(-2.5, r2, -15.0 * r1)) * (id2 * id2);
                        double t = -g / h;
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            const double p = fma(t, fma(t, fma(t, fma(t, q4, q3), q2), h), g);
                            const double dp = fma(t, fma(t, fma(4.0 * t, q4, 3.0 * q3), 2.0 * q2), h);
                            t = (dp < 0.0) ? t - p / dp : t;
                        }
                        // step, span, the interpolant's error term t^2 |d|^3, higher-order terms against h, rounding of the values
                        if (fabs(t) < kHermiteTol && fabs(t) < 0.5 * fabs(d) && fabs(t * d) < kHermiteSpan &&
                            t * t * fabs(d * d * d) < kHermiteT2D3 &&
                            fabs(t * fma(t, fma(t, q4, q3), q2)) < kHermiteGuard * fabs(h) &&
                            kHermiteNoise * fabs(f) * (t * t) < fabs(h) * fabs(d * d * d)) {
                            f = fma(t, fma(t, fma(t, fma(t, fma(t, 0.2 * q4, 0.25 * q3), q2 * (1.0 / 3.0)), 0.5 * h), g), f);
                            un = u + t;
                            hermite = true;
                            flg = TPHIP_FLAG_OK;
                            if (un >= kUCheck) { checking = true; u_conv = un; f_conv = f; un = kUMax; flg = -1; }
                        }
                    }
                    if (!hermite) {
                    // Concave: Newton's -g/h refined to log(1 - g/h), the exact maximiser of m u - a exp(u) + c
                    // fitted to (g, h) (same rule as the oracle); convex: a capped step uphill.
                    double step;
                    if (h < 0.0) {
                        const double q = 1.0 - g / h;
                        step = (q > 0.0) ? log(q) : -g / h;
                        // First step from the parsimony start: as u -> -inf the slope of log L tends to the column's parsimony
                        // length m, which classify_kernel counted.  Fitting m u - a exp(b u) + c to (m, g, h) -- one more shape
                        // parameter than the model above, whose m is implied by g - h -- puts the first step nearer the optimum
                        // (median miss 4e-4 instead of 7.5e-4 log-units: more columns leave after two evaluations; same rule as
                        // the oracle): (1/b) log(m / (m - g)), b = -h / (m - g).
                        if (!have_prev && h_prev > 0.0) {
                            const double A = h_prev - g;
                            const double sb = (A > 0.0) ? log(h_prev / A) * (A / -h) : kStepMax + 1.0;
                            if (fabs(sb) <= kStepMax) step = sb;
                        }
                    } else {
                        step = uphill ? kStepMax : -kStepMax;
                    }
                    // Weakly curved concave point with a previous concave point nearby: Halley's step with the third derivative
                    // from the last two curvatures (same rule as the oracle, which says why): on the approach to a flat maximum
                    // at a large rate Newton-type steps converge linearly, and in a small batch the launch waits for that lane.
                    if (have_prev && h < 0.0 && h_prev < 0.0 && !regular_here && fabs(u - u_prev) < kHalleySpan) {
                        const double f3h = (h - h_prev) / (u - u_prev);
                        const double den = h - 0.5 * g * f3h / h;
                        if (den < 0.0) {
                            const double sh = -g / den;
                            if (fabs(sh) <= kStepMax && (sh > 0.0) == (g > 0.0)) step = sh;
                        }
                    }
                    if (!(step <= kStepMax)) step = kStepMax;
                    if (step < -kStepMax) step = -kStepMax;
                    // Plateau stride (same rule as the oracle): still uphill at a rate of 20 or more with nothing known
                    // above is almost always a column whose log L creeps up to its s -> infinity asymptote; g and h shrink
                    // together there and the steps stay at ~0.1 for 20-30 evaluations until the flatness rule fires --
                    // and in a small batch one such lane keeps its whole wave alive.  A stride of at least
                    // kPlateauStride covers the two log-units to flatness in a few evaluations; a maximum that does lie
                    // ahead is overshot by at most that much and then bracketed from both sides.
                    if (uphill && hi_open && u >= kUCheck && step < kPlateauStride) step = kPlateauStride;
                    const double tol = (have_prev && regular_here) ? kStepTol : kStepTolFirst;
                    un = u + step;
                    // the bracket safeguard must not see a converged (possibly underflowing) step
                    if (fabs(step) >= tol) {
                        if (un >= hi) un = hi_open ? kUMax : 0.5 * (lo + hi);
                        else if (un <= lo) un = lo_open ? kUMin : 0.5 * (lo + hi);
                        step = un - u;
                    }
                    // Converged: the last step gets its third-order correction (third derivative f3 from the two most
                    // recent curvatures) instead of one more evaluation -- same rule as the oracle.  Where that
                    // correction is a sizeable part of the step itself (weak curvature, far previous point) the model
                    // behind it is not good enough to stop on: iterate once more unless the step is already negligible.
                    double f3 = h, corr = 0.0;
                    if (have_prev && h < 0.0) {
                        f3 = (h - h_prev) / (u - u_prev);
                        corr = 0.5 * ((f3 - h) / h) * step * step;
                    }
                    if (fabs(step) < tol && (fabs(corr) <= 0.05 * fabs(step) || fabs(step) < kStepTolFirst)) {
                        step -= corr;
                        f = fma(step, fma(step, fma(step, f3 / 6.0, 0.5 * h), g), f);
                        un = u + step;
                        flg = TPHIP_FLAG_OK;
                        if (un >= kUCheck) {   // confirm by value before believing a maximum this far out
                            checking = true; u_conv = un; f_conv = f; un = kUMax; flg = -1;
                        }
                    }
                    }
                    u_prev = u; h_prev = h; g_prev = g; have_prev = true;
                    f_conv = checking ? f_conv : f_eval;   // previous value for the two-point model (see above)
                    u = un;
                    if (flg < 0 && it >= kMaxIt) flg = TPHIP_FLAG_MAXIT;
                }
                if (flg >= 0) {  // this column is finished: write it out, the lane becomes free
                    if constexpr (MIXED) kappa = reinterpret_cast<const double*>(P.models + seg_locus[myk])[31];
                    const double r = exp(u) * kappa;
                    P.rate[col] = r;
                    P.subst[col] = r * P.chrono_length;
                    P.lnl[col] = f;
                    P.flag[col] = (uint8_t)flg;
                    done = true;
                }
            }
            // lane refill: free lanes take the next columns of the segment, in lane order
            const unsigned long long free_mask = __ballot(done);
            if (free_mask == 0ull) continue;
            if (next < end) {
                const int rank = __popcll(free_mask & ((1ull << lane) - 1ull));
                const int idx = next + rank;
                const int32_t picked = __shfl(cand, rank);   // = work[next + rank]
                if (done && idx < end) {
                    col = MIXED ? (picked & 0x0fffffff) : picked;
                    if constexpr (MIXED) {
                        const int nk = picked >> 28;
                        if (nk != myk) {   // the lane moves on to another locus: its model and tip table
                            myk = nk;
                            load_model_lane(P.models + seg_locus[nk], R);
                            wt = wtab + 64 * nk;
                        }
                    }
#pragma unroll
                    for (int w = 0; w < NW; ++w) pk[w] = (w < P.nwords) ? P.packed[(int64_t)w * P.ncols_total + col] : 0u;
                    if constexpr (NW == kStreamWords) pk[0] = P.packed[col];
                    u = P.rate[col]; h_prev = P.subst[col]; lo = kUMin; hi = kUMax; lo_open = true; hi_open = true; have_prev = false; it = 0;
                    checking = false;
                    done = false;
                }
                next += __popcll(free_mask);
                if (next < end) cand = fetch(next + lane < end ? next + lane : end - 1);
            } else if (free_mask == ~0ull) {
                break;  // segment exhausted and every lane finished
            }
        }
    }
    // evaluation count for the FLOP model (one atomic per wave)
    unsigned tot = evals;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) tot += __shfl_xor(tot, o);
    if (lane == 0 && P.eval_counter) atomicAdd(P.eval_counter, (unsigned long long)tot);
#ifdef TPHIP_SITE_TRACE_ROUNDS   // diagnostic build: rounds and wall-clock ticks (100 MHz) per wave, sum and maximum
    if (lane == 0 && P.eval_counter) {
        const unsigned long long ticks = wall_clock64() - tick0;
        atomicAdd(P.eval_counter + 1, (unsigned long long)rounds);
        atomicMax(P.eval_counter + 2, (unsigned long long)rounds);
        atomicAdd(P.eval_counter + 3, ticks);
        atomicMax(P.eval_counter + 4, ticks);
        atomicAdd(P.eval_counter + 5, 1ull);
    }
#endif
}

// Diagnostic: evaluate f = log L, g = df/du, h = d2f/du2 at a caller-chosen u for EVERY column (no
// classification, no optimiser).  Tests use it to check the derivative propagation by finite differences.
// Uses the same chunk tables with chunk_cols columns per workgroup, 64 at a time.
__global__ __launch_bounds__(kSiteBlock) void eval_columns_kernel(EvalParams E) {
    extern __shared__ double lds[];
    double* wtab = lds;
    double* mtab = lds + 64;
    double* etab = lds + 96;
    double* stack = lds + kSiteLdsHeader;
    etab[threadIdx.x] = kExp2Table[threadIdx.x];
    const SiteParams& P = E.S;
    const int chunk = blockIdx.x;
    const int locus = P.chunk_locus[chunk];
    const LocusModel* __restrict__ M = P.models + locus;
    const int lane = threadIdx.x;
    build_tip_table(M, wtab, lane);
    const ModelRegs R = load_model(M, mtab, lane);
    const int64_t lo = P.locus_offsets[locus], hi = P.locus_offsets[locus + 1];
    const int64_t S = hi - lo;
    int64_t ns = (S + P.chunk_cols / 2) / P.chunk_cols;  // = workgroups launched for this locus
    ns = ns < 1 ? 1 : ns;
    const int64_t j = P.chunk_index[chunk];
    const int64_t first = lo + j * S / ns;
    const int64_t last = lo + (j + 1) * S / ns;
    for (int64_t base = first; base < last; base += kSiteBlock) {
        const int64_t want = base + lane;
        const bool active = want < last;
        const int64_t col = active ? want : first;
        double f, g, h;
        const uint32_t none[1] = {0};
        evaluate_site<0>(P, R, wtab, etab, stack, col, none, exp(E.u[col]), f, g, h);
        if (active) { E.f[col] = f; E.g[col] = g; E.h[col] = h; }
    }
}

}  // namespace tphip
