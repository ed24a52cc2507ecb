// fast_exp.hpp -- exp(x) for x <= 0 on the FP64 VALU, shared by the site-rate and PI kernels.
#pragma once
#include <hip/hip_runtime.h>

namespace tphip {

// exp(x) for x <= 0 (branch exponents lam*t*s are never positive).
//   n = round(x*log2 e) by the 1.5*2^52 shift trick (one FMA leaves n in the low mantissa bits, one ADD gives it
//   back as a double: no v_rndne / v_cvt), Cody-Waite reduction r = x - n ln2 (two FMAs), degree-13 Taylor
//   polynomial on |r| <= ln2/2 (truncation < 4e-18), 2^n applied by adding n to the exponent field.
// x is clamped at -708 (exp(-708) = 3e-308 is the smallest normal result; anything smaller contributes nothing
// to a likelihood here).  Error vs libm <= 1 ulp on [-708, 0].
__device__ __forceinline__ double exp_nonpos(double x) {
    const double LOG2E = 1.4426950408889634074;
    const double LN2_HI = 6.93147180369123816490e-01;
    const double LN2_LO = 1.90821492927058770002e-10;
    const double SHIFT = 6755399441055744.0;  // 1.5 * 2^52
    x = fmax(x, -708.0);
    const double t = fma(x, LOG2E, SHIFT);    // low 32 bits of t = n (two's complement)
    const double n = t - SHIFT;
    double r = fma(-n, LN2_HI, x);
    r = fma(-n, LN2_LO, r);
    double p = 1.6059043836821613e-10;          // 1/13!
    p = fma(p, r, 2.08767569878681e-09);        // 1/12!
    p = fma(p, r, 2.505210838544172e-08);       // 1/11!
    p = fma(p, r, 2.755731922398589e-07);       // 1/10!
    p = fma(p, r, 2.7557319223985893e-06);      // 1/9!
    p = fma(p, r, 2.48015873015873e-05);        // 1/8!
    p = fma(p, r, 1.984126984126984e-04);       // 1/7!
    p = fma(p, r, 1.388888888888889e-03);       // 1/6!
    p = fma(p, r, 8.333333333333333e-03);       // 1/5!
    p = fma(p, r, 4.1666666666666664e-02);      // 1/4!
    p = fma(p, r, 1.6666666666666666e-01);      // 1/3!
    p = fma(p, r, 0.5);
    p = fma(p, r, 1.0);
    p = fma(p, r, 1.0);
    // p in [0.70, 1.42]; n >= -1021 after the clamp, so adding n to the exponent field keeps the result normal
    const int ni = (int)(unsigned)__double_as_longlong(t);
    const int hi = __double2hiint(p) + (int)((unsigned)ni << 20);
    return __hiloint2double(hi, __double2loint(p));
}

// 2^(j/64), j = 0..63, correctly rounded; copied into LDS per workgroup for exp_nonpos_tab.
__device__ __constant__ const double kExp2Table[64] = {
    1.0, 1.0108892860517005, 1.0218971486541166, 1.0330248790212284,
    1.0442737824274138, 1.0556451783605572, 1.0671404006768237, 1.0787607977571199,
    1.0905077326652577, 1.102382583307841, 1.1143867425958924, 1.1265216186082418,
    1.1387886347566916, 1.1511892299529827, 1.1637248587775775, 1.1763969916502812,
    1.189207115002721, 1.202156731452703, 1.215247359980469, 1.22848053610687,
    1.241857812073484, 1.255380757024691, 1.2690509571917332, 1.2828700160787783,
    1.2968395546510096, 1.3109612115247644, 1.3252366431597413, 1.339667524053303,
    1.3542555469368927, 1.3690024229745905, 1.383909881963832, 1.3989796725383112,
    1.4142135623730951, 1.42961333839197, 1.4451808069770467, 1.460917794180647,
    1.4768261459394993, 1.4929077282912648, 1.5091644275934228, 1.5255981507445384,
    1.5422108254079407, 1.559004400237837, 1.5759808451078865, 1.593142151342267,
    1.6104903319492543, 1.6280274218573478, 1.645755478153965, 1.6636765803267364,
    1.681792830507429, 1.7001063537185235, 1.718619298122478, 1.7373338352737062,
    1.7562521603732995, 1.7753764925265212, 1.7947090750031072, 1.8142521755003989,
    1.8340080864093424, 1.8539791250833855, 1.8741676341103, 1.8945759815869656,
    1.9152065613971474, 1.9360617934922943, 1.9571441241754002, 1.978456026387951};

// Table-driven exp(x), x <= 0 (Tang 1989 with a 64-entry table): x = (64 n + j) ln2/64 + r, |r| <= ln2/128,
// exp(x) = 2^n * T[j] * (1 + q(r)), q of degree 5 (truncation r^6/720 < 4e-17).  11 FP64 ops + one LDS read
// instead of 19 FP64 ops; error <= 1.5 ulp.
// c4: the polynomial's coefficient 1/24 held in a VGPR by the caller.  fma(r, 1/120, 1/24) has two constant operands and a
// VOP3 instruction takes one from the scalar side, so the compiler re-materialised the other with a v_mov_b64 in front of
// every exp (3 per tree op); a register the compiler cannot see through (empty asm, see site_rate_kernel) stays put.
__device__ __forceinline__ double exp_nonpos_tab(double x, const double* __restrict__ etab, double c4 = 4.1666666666666664e-02) {
    const double INV = 92.33248261689366;           // 64 / ln2
    const double L_HI = 0.01083042469326756;        // ln2/64, 32 significant bits: k*L_HI is exact for |k| < 2^20
    const double L_LO = 2.9815858269852933e-12;
    const double SHIFT = 6755399441055744.0;        // 1.5 * 2^52
    // (no clamp at -708: 2^n is applied with ldexp, which underflows to 0 by itself, and the shift trick holds for
    //  |x| < 2^24 / INV * 2^7 -- site rates stop at s = 1e4, so |x| = |lambda t s| stays below ~1e5)
    const double t = fma(x, INV, SHIFT);            // low 32 bits of t = k = 64 n + j
    const double kd = t - SHIFT;
    double r = fma(-kd, L_HI, x);
    r = fma(-kd, L_LO, r);
    const int k = (int)(unsigned)__double_as_longlong(t);
    const double T = etab[k & 63];
    double q = fma(r, 8.333333333333333e-03, c4);
    q = fma(q, r, 1.6666666666666666e-01);
    q = fma(q, r, 0.5);
    q = fma(q, r, 1.0);
    q = q * r;                                      // q = exp(r) - 1
    const double p = fma(T, q, T);                  // in [0.99, 2.0)
    // 2^n through v_ldexp_f64 (shift + ldexp) rather than integer arithmetic on the exponent field (shift, mask,
    // add): same bits (n >= -1022 after the clamp), one instruction fewer per exp, C3 site rates -1 % (A/B on one box)
    return ldexp(p, k >> 6);
}

}  // namespace tphip
