// erm_rng.hpp -- device-side counter-based streams and scalar samplers (gfx950).
//
// Sampling specification (shared with the CPU oracle, which restates it independently in C):
//   stream(site, i, j, sweep) = words of Philox4x32-10(key = seed; ctr = {i, j, sweep, site<<24 | chain<<16 | k}),
//   k = 0,1,2,..., consumed strictly in order.  Because draws are addressed by (site, i, j, sweep) and not by
//   thread id, results do not depend on launch geometry.
//   uniform: fp64 (x + 1/2) 2^-32; fp32 ((x >> 9) + 1/2) 2^-23 (the same value truncated to 23 bits: every value is exactly
//            representable in fp32, lies strictly inside (0,1) and never makes cospi(2u) vanish)
//   expo   : -log(u);  normal: sqrt(-2 log u1) cos(2 pi u2) (two words per variate)
//   PG(1,c): Polson-Scott-Windle / Devroye alternating-series sampler, t = 0.64, as a single-level rejection sampler
//            with one Philox block per attempt (replaces PolyaGammaPSWSampler(1, eta) at /root/reference/src/Draw.pl.jl:38)
//   IG     : Michael-Schucany-Haas (replaces Distributions.InverseGaussian at src/Draw.pl.jl:312,335)
//   TN, Gamma (fp64 only; item-level draws): Robert (1995) / Marsaglia-Tsang (2000)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "erm_layout.hpp"

namespace erm {

constexpr int MAX_TRIES = 4096;   // bound on every rejection loop: a non-finite input yields NaN (caught by the tiny step's check), never a hang

enum Site : uint32_t {
    SITE_OMEGA = 1, SITE_THETA = 2, SITE_ZETA = 3, SITE_NU = 4, SITE_B = 5, SITE_A = 6,
    SITE_LAMBDA = 7, SITE_SIG2T = 8, SITE_BETA = 9, SITE_SIGP = 10, SITE_RHO = 11, SITE_DATA_SUBJ = 12, SITE_DATA_CELL = 13, SITE_TEST = 15
};

// ---- precision-generic math wrappers -------------------------------------------------------
// fp32 uses the hardware-rate transcendental instructions (v_exp_f32, v_log_f32, v_rcp_f32, v_sqrt_f32, v_cos_f32: ~1 ulp),
// never the IEEE-exact expansion sequences: the cell path is VALU-issue-bound and these are its inner loop.
__device__ __forceinline__ float  r_exp(float x)  { return __builtin_amdgcn_exp2f(x * 1.44269504088896341f); }
__device__ __forceinline__ double r_exp(double x) { return exp(x); }
__device__ __forceinline__ float  r_log(float x)  { return __builtin_amdgcn_logf(x) * 0.693147180559945309f; }
__device__ __forceinline__ double r_log(double x) { return log(x); }
__device__ __forceinline__ float  r_sqrt(float x)  { return __builtin_amdgcn_sqrtf(x); }
__device__ __forceinline__ double r_sqrt(double x) { return sqrt(x); }
__device__ __forceinline__ float  r_rcp(float x)  { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ double r_rcp(double x) { return 1.0 / x; }
__device__ __forceinline__ float  r_div(float a, float b)  { return a * __builtin_amdgcn_rcpf(b); }
__device__ __forceinline__ double r_div(double a, double b) { return a / b; }
__device__ __forceinline__ float  r_abs(float x)  { return fabsf(x); }
__device__ __forceinline__ double r_abs(double x) { return fabs(x); }
__device__ __forceinline__ float  r_cos2pi(float x)  { return __builtin_amdgcn_cosf(x); }     // v_cos_f32 takes revolutions
__device__ __forceinline__ double r_cos2pi(double x) { return cospi(2.0 * x); }

// ---- fp64 elementary functions for the cell path ---------------------------------------------------
// The OCML routines are correctly-rounded-grade and guard every special case: log(double) is 98 VALU instructions (double-double
// arithmetic), log1p 135, sqrt 22, exp 42.  The cell path calls them on arguments whose range it knows (uniforms in (0,1), products of
// them, exponents in [0, 700]) and needs ~1e-15 relative accuracy, not the last bit: these forms take 33 / 10 / 19 instructions.
namespace fm {
__device__ __forceinline__ double rcp(double b)                   // 1/b, b normal and finite: hardware estimate + two Newton steps
{
    double r = __builtin_amdgcn_rcp(b);
    double e = fma(-b, r, 1.0); r = fma(r, e, r);
    e = fma(-b, r, 1.0); r = fma(r, e, r);
    return r;
}
__device__ __forceinline__ double div(double a, double b)         // a/b to < 1 ulp for normal, finite operands and quotient
{
    const double r = rcp(b), q = a * r;
    return fma(fma(-b, q, a), r, q);
}
__device__ __forceinline__ double sqrt(double a)                  // a > 0 normal: rsq estimate, one coupled Newton step, two corrections
{
    const double y = __builtin_amdgcn_rsq(a);
    double g = a * y, h = 0.5 * y;
    const double r = fma(-h, g, 0.5);
    g = fma(g, r, g); h = fma(h, r, h);
    g = fma(fma(-g, g, a), h, g);
    g = fma(fma(-g, g, a), h, g);
    return g;
}
// log(x), x > 0 normal: x = m 2^e with m in [sqrt(1/2), sqrt(2)), s = (m-1)/(m+1), log m = 2 atanh(s) = 2s (1 + s^2/3 + s^4/5 + ...);
// |s| <= 0.1716, so the series is cut after s^20/21 (2.3e-17)
__device__ __forceinline__ double log(double x)
{
    double m = __builtin_amdgcn_frexp_mant(x);                    // [1/2, 1)
    int e = __builtin_amdgcn_frexp_exp(x);
    const bool lo = m < 0.70710678118654752440;
    m = __builtin_amdgcn_ldexp(m, lo ? 1 : 0); e -= lo ? 1 : 0;
    const double f = m - 1.0;
    const double s = div(f, 2.0 + f), z = s * s;
    double P = 1.0 / 21.0;
    P = fma(P, z, 1.0 / 19.0); P = fma(P, z, 1.0 / 17.0); P = fma(P, z, 1.0 / 15.0); P = fma(P, z, 1.0 / 13.0); P = fma(P, z, 1.0 / 11.0);
    P = fma(P, z, 1.0 / 9.0); P = fma(P, z, 1.0 / 7.0); P = fma(P, z, 1.0 / 5.0); P = fma(P, z, 1.0 / 3.0);
    const double s2 = s + s;
    const double lm = fma(s2, P * z, s2);
    return fma((double)e, 0.693147180559945309417, lm);
}
// log(x) with a 128-entry table in LDS (fill_log_table): x = m 2^e, m in [1, 2), k = the top 7 fraction bits of m, c_k = 1 + (k + 1/2)/128,
// tab[k] = (1/c_k, log c_k); r = m/c_k - 1 (one fma, |r| <= 2^-8) and log x = e ln2 + log c_k + log1p(r), log1p by 6 terms (next: r^7/7 < 2e-18).
// 17 instructions with a 7-deep dependent chain, against 33 with a 25-deep one for the table-free form; absolute error ~2e-16 (the relative
// error near x = 1 is larger: every caller adds the result to, or takes the root of, something of order one)
__device__ __forceinline__ void fill_log_table(double2* tab, int tid, int nthreads)
{
    for (int k = tid; k < 128; k += nthreads) { const double c = 1.0 + ((double)k + 0.5) * (1.0 / 128.0); tab[k] = make_double2(1.0 / c, ::log(c)); }
}
__device__ __forceinline__ double log(double x, const double2* tab)
{
    const uint32_t hi = (uint32_t)__double2hiint(x), lo = (uint32_t)__double2loint(x);
    const int e = (int)((hi >> 20) & 0x7FFu) - 1023;
    const double m = __hiloint2double((int)((hi & 0x000FFFFFu) | 0x3FF00000u), (int)lo);
    const double2 t = tab[(hi >> 13) & 0x7Fu];
    const double r = fma(m, t.x, -1.0);
    double P = fma(r, -1.0 / 6.0, 0.2);
    P = fma(P, r, -0.25); P = fma(P, r, 1.0 / 3.0); P = fma(P, r, -0.5);
    const double l1p = fma(P * r, r, r);
    return fma((double)e, 0.693147180559945309417, t.y) + l1p;
}
// log((w + 1/2) 2^-32), the logarithm of the uniform a 32-bit word stands for (word_to_unif<double>): w + 1/2 is exact in fp64 and the scaling only moves the
// exponent, so the table form runs on w + 1/2 and takes the 32 off the exponent it extracts -- the same bits as log(word_to_unif<double>(w), tab), one v_ldexp_f64 less
__device__ __forceinline__ double log_word(uint32_t w, const double2* tab)
{
    const double x = (double)w + 0.5;
    const uint32_t hi = (uint32_t)__double2hiint(x), lo = (uint32_t)__double2loint(x);
    const int e = (int)((hi >> 20) & 0x7FFu) - (1023 + 32);
    const double m = __hiloint2double((int)((hi & 0x000FFFFFu) | 0x3FF00000u), (int)lo);
    const double2 t = tab[(hi >> 13) & 0x7Fu];
    const double r = fma(m, t.x, -1.0);
    double P = fma(r, -1.0 / 6.0, 0.2);
    P = fma(P, r, -0.25); P = fma(P, r, 1.0 / 3.0); P = fma(P, r, -0.5);
    const double l1p = fma(P * r, r, r);
    return fma((double)e, 0.693147180559945309417, t.y) + l1p;
}
// cos(2 pi u), u in [0, 1]: u = k/4 + r with k = rint(4u) and |r| <= 1/8 revolutions, i.e. phi = 2 pi r in [-pi/4, pi/4];
// cos(2 pi u) = cos(phi + k pi/2) = {cos, -sin, -cos, sin}(phi) for k mod 4 = 0..3; Taylor polynomials to phi^16 / phi^15 (next terms < 5e-17).
// ~30 instructions against 66 for OCML's cospi.
__device__ __forceinline__ double cos2pi(double u)
{
    const double k = __builtin_rint(4.0 * u);
    const double phi = 6.28318530717958647692 * fma(-0.25, k, u);        // exact reduction: 4u and k are exact for u = (w + 1/2) 2^-32
    const double z = phi * phi;
    double c = 1.0 / 20922789888000.0;                                    // cos: sum (-1)^n z^n / (2n)!
    c = fma(c, z, -1.0 / 87178291200.0); c = fma(c, z, 1.0 / 479001600.0); c = fma(c, z, -1.0 / 3628800.0); c = fma(c, z, 1.0 / 40320.0);
    c = fma(c, z, -1.0 / 720.0); c = fma(c, z, 1.0 / 24.0); c = fma(c, z, -0.5); c = fma(c, z, 1.0);
    double sn = -1.0 / 1307674368000.0;                                   // sin: phi sum (-1)^n z^n / (2n+1)!
    sn = fma(sn, z, 1.0 / 6227020800.0); sn = fma(sn, z, -1.0 / 39916800.0); sn = fma(sn, z, 1.0 / 362880.0); sn = fma(sn, z, -1.0 / 5040.0);
    sn = fma(sn, z, 1.0 / 120.0); sn = fma(sn, z, -1.0 / 6.0); sn = fma(sn * z, phi, phi);
    const int q = (int)k & 3;
    const double v = (q & 1) ? sn : c;
    return (q == 0 || q == 3) ? v : -v;
}
// e^{-a}, 0 <= a (-> 0 beyond 700): a = k ln2 + r, |r| <= ln2/2, Taylor series of e^{-r} to r^13/13! (4e-18), scaled by 2^-k
__device__ __forceinline__ double exp_neg(double a)
{
    a = a < 700.0 ? a : 700.0;
    const double k = __builtin_rint(a * -1.44269504088896340736);     // -rint(a / ln2): the negated count goes straight into the reduction and the final scaling
    double r = fma(k, 0.693147180559945286227, a);                // ln2 = hi + lo with hi = double(ln2): the fma forms a - |k| hi exactly
    r = fma(k, 2.31904681384629955842e-17, r);
    const double t = -r;
    double P = 1.0 / 6227020800.0;
    P = fma(P, t, 1.0 / 479001600.0); P = fma(P, t, 1.0 / 39916800.0); P = fma(P, t, 1.0 / 3628800.0); P = fma(P, t, 1.0 / 362880.0);
    P = fma(P, t, 1.0 / 40320.0); P = fma(P, t, 1.0 / 5040.0); P = fma(P, t, 1.0 / 720.0); P = fma(P, t, 1.0 / 120.0); P = fma(P, t, 1.0 / 24.0);
    P = fma(P, t, 1.0 / 6.0); P = fma(P, t, 0.5); P = fma(P, t, 1.0); P = fma(P, t, 1.0);
    return __builtin_amdgcn_ldexp(P, (int)k);
}
// The same function for the cell log-likelihood's factors 1 + e^{-|eta|} (five million per sweep; the samplers' decisions keep exp_neg above): the degree-11 polynomial
// that interpolates e^t at the Chebyshev nodes of |t| <= 0.3466 -- 1.2e-17 absolute, against 4e-18 for the Taylor series cut after t^13 and 1.1e-16 for the rounding of
// the result -- two multiply-adds fewer per cell (coefficients: tools/exp_poly.py, computed with 60 digits).
__device__ __forceinline__ double exp_neg_ll(double a)
{
    a = a < 700.0 ? a : 700.0;
    const double k = __builtin_rint(a * -1.44269504088896340736);
    const double r = fma(k, 0.693147180559945286227, a);          // (without the low word of ln 2: it moves e^{-a} by |k| 2.3e-17 relative, i.e. the factor 1 + e^{-a} by < 1.2e-17)
    const double t = -r;
    double P = 0x1.af6326f3df789p-26;
    P = fma(P, t, 0x1.28b40d95cf927p-22); P = fma(P, t, 0x1.71ddf56f3c074p-19); P = fma(P, t, 0x1.a01991a3c8c2ep-16); P = fma(P, t, 0x1.a01a01b1457b9p-13);
    P = fma(P, t, 0x1.6c16c187ff24ap-10); P = fma(P, t, 0x1.111111110f220p-7); P = fma(P, t, 0x1.555555554f0bfp-5); P = fma(P, t, 0x1.555555555555ap-3);
    P = fma(P, t, 0x1.0000000000011p-1); P = fma(P, t, 1.0); P = fma(P, t, 1.0);
    return __builtin_amdgcn_ldexp(P, (int)k);
}
}  // namespace fm

// "quick" forms for operands known to be normal, finite (and positive where a root or logarithm is taken): fp32 = the hardware-rate instruction,
// fp64 = the range-specialised forms above (<= 1 ulp) instead of the IEEE division (30 instructions), OCML's sqrt (22), log (98) and cospi (66)
__device__ __forceinline__ float  q_rcp(float x)  { return r_rcp(x); }
__device__ __forceinline__ double q_rcp(double x) { return fm::rcp(x); }
__device__ __forceinline__ float  q_div(float a, float b)  { return r_div(a, b); }
__device__ __forceinline__ double q_div(double a, double b) { return fm::div(a, b); }
__device__ __forceinline__ float  q_sqrt(float x)  { return r_sqrt(x); }
__device__ __forceinline__ double q_sqrt(double x) { return fm::sqrt(x); }
__device__ __forceinline__ float  q_log(float x)  { return r_log(x); }
__device__ __forceinline__ double q_log(double x) { return fm::log(x); }
__device__ __forceinline__ float  q_cos2pi(float x)  { return r_cos2pi(x); }
__device__ __forceinline__ double q_cos2pi(double x) { return fm::cos2pi(x); }
__device__ __forceinline__ double q_exp_neg(double a) { return fm::exp_neg(a); }       // e^{-a}, a >= 0

template <typename real> struct Const;
template <> struct Const<float> {
    static constexpr float PI = 3.14159265358979323846f;
    static constexpr float SQRT1_2 = 0.70710678118654752440f;
};
template <> struct Const<double> {
    static constexpr double PI = 3.14159265358979323846;
    static constexpr double SQRT1_2 = 0.70710678118654752440;
};

// ---- Philox4x32-10 --------------------------------------------------------------------------
__device__ __forceinline__ void philox_round(uint32_t& c0, uint32_t& c1, uint32_t& c2, uint32_t& c3, uint32_t k0, uint32_t k1)
{
    const uint64_t p0 = (uint64_t)0xD2511F53u * c0;
    const uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
    // one three-input XOR per word (v_bitop3_b32, truth table 0x96) instead of the two two-input ones the compiler emits
    uint32_t n0, n2;
    asm("v_bitop3_b32 %0, %1, %2, %3 bitop3:0x96" : "=v"(n0) : "v"((uint32_t)(p1 >> 32)), "v"(c1), "s"(k0));
    asm("v_bitop3_b32 %0, %1, %2, %3 bitop3:0x96" : "=v"(n2) : "v"((uint32_t)(p0 >> 32)), "v"(c3), "s"(k1));
    const uint32_t n1 = (uint32_t)p1;
    const uint32_t n3 = (uint32_t)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
}

__device__ __forceinline__ void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1,
                                              uint32_t& o0, uint32_t& o1, uint32_t& o2, uint32_t& o3)
{
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        philox_round(c0, c1, c2, c3, k0, k1);
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    o0 = c0; o1 = c1; o2 = c2; o3 = c3;
}

// The same block with its first NV round keys held in VECTOR registers (kv[], filled once by philox_vector_keys) and the rest derived from (k0, k1) as
// scalars.  The cell loop of the row pass is short of scalar registers: the compiler parks round keys in VGPR lanes and fetches each with a
// v_readlane_b32 -- a VALU instruction -- every trip (9 of the loop's 174); spare vector registers hold them for nothing.
template <int NV>
__device__ __forceinline__ void philox_vector_keys(uint32_t k0, uint32_t k1, uint32_t (&kv)[NV > 0 ? NV : 1])
{
#pragma unroll
    for (int e = 0; e < NV; ++e) {
        const uint32_t k = (e & 1) ? k1 + (uint32_t)(e >> 1) * 0xBB67AE85u : k0 + (uint32_t)(e >> 1) * 0x9E3779B9u;
        asm volatile("v_mov_b32 %0, %1" : "=v"(kv[e]) : "s"(k));      // opaque: the compiler must not turn the copy back into a scalar
    }
}
template <int NV>
__device__ __forceinline__ void philox4x32_10_vk(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1, const uint32_t (&kv)[NV > 0 ? NV : 1],
                                                 uint32_t& o0, uint32_t& o1, uint32_t& o2, uint32_t& o3)
{
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        const uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t n0, n2;
        if (2 * r < NV) asm("v_bitop3_b32 %0, %1, %2, %3 bitop3:0x96" : "=v"(n0) : "v"((uint32_t)(p1 >> 32)), "v"(c1), "v"(kv[2 * r < NV ? 2 * r : 0]));
        else asm("v_bitop3_b32 %0, %1, %2, %3 bitop3:0x96" : "=v"(n0) : "v"((uint32_t)(p1 >> 32)), "v"(c1), "s"(k0));
        if (2 * r + 1 < NV) asm("v_bitop3_b32 %0, %1, %2, %3 bitop3:0x96" : "=v"(n2) : "v"((uint32_t)(p0 >> 32)), "v"(c3), "v"(kv[2 * r + 1 < NV ? 2 * r + 1 : 0]));
        else asm("v_bitop3_b32 %0, %1, %2, %3 bitop3:0x96" : "=v"(n2) : "v"((uint32_t)(p0 >> 32)), "v"(c3), "s"(k1));
        c1 = (uint32_t)p1; c3 = (uint32_t)p0; c0 = n0; c2 = n2;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    o0 = c0; o1 = c1; o2 = c2; o3 = c3;
}

// The same block for the cell streams of the row pass, counter (i, j, c2, S | attempt): rounds 0 and 1 multiply words that do not depend on the attempt --
// M0 i (the subject), M1 c2 (uniform) and M0 (hi(M1 c2) ^ j ^ k0) (the item, once per sweep: philox_item_product) -- so a cell fixes two of round 1's four output
// words when it is taken from the queue (m2 = hi(item product) ^ lo(M0 i) ^ (k1 + G1), m3 = lo(item product)) and an attempt spends ONE 32 x 32 -> 64 multiplication on
// rounds 0-1 instead of three (v_mad_u64_u32 issues in 9 cycles: profiles/round4_valu_issue_rates.txt).  s_n1k = lo(M1 c2) ^ (k0 + G0).  Same words, bit for bit.
__device__ __forceinline__ uint2 philox_item_product(uint32_t j, uint32_t c2, uint32_t k0)
{
    const uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ j ^ k0;
    const uint64_t pj = (uint64_t)0xD2511F53u * n0;
    return make_uint2((uint32_t)(pj >> 32), (uint32_t)pj);
}
template <int NV>
__device__ __forceinline__ void philox4x32_10_vk_cell(uint32_t p0hi, uint32_t m2, uint32_t m3, uint32_t c3w, uint32_t s_n1k, uint32_t k0, uint32_t k1,
                                                      const uint32_t (&kv)[NV > 0 ? NV : 1], uint32_t& o0, uint32_t& o1, uint32_t& o2, uint32_t& o3)
{
    uint32_t n2;
    if (1 < NV) asm("v_bitop3_b32 %0, %1, %2, %3 bitop3:0x96" : "=v"(n2) : "v"(p0hi), "v"(c3w), "v"(kv[1 < NV ? 1 : 0]));
    else asm("v_bitop3_b32 %0, %1, %2, %3 bitop3:0x96" : "=v"(n2) : "v"(p0hi), "v"(c3w), "s"(k1));
    const uint64_t q1 = (uint64_t)0xCD9E8D57u * n2;
    uint32_t c0 = (uint32_t)(q1 >> 32) ^ s_n1k, c1 = (uint32_t)q1, c2 = m2, c3 = m3;
    k0 += 2u * 0x9E3779B9u; k1 += 2u * 0xBB67AE85u;
#pragma unroll
    for (int r = 2; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        const uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t n0, n2_;
        if (2 * r < NV) asm("v_bitop3_b32 %0, %1, %2, %3 bitop3:0x96" : "=v"(n0) : "v"((uint32_t)(p1 >> 32)), "v"(c1), "v"(kv[2 * r < NV ? 2 * r : 0]));
        else asm("v_bitop3_b32 %0, %1, %2, %3 bitop3:0x96" : "=v"(n0) : "v"((uint32_t)(p1 >> 32)), "v"(c1), "s"(k0));
        if (2 * r + 1 < NV) asm("v_bitop3_b32 %0, %1, %2, %3 bitop3:0x96" : "=v"(n2_) : "v"((uint32_t)(p0 >> 32)), "v"(c3), "v"(kv[2 * r + 1 < NV ? 2 * r + 1 : 0]));
        else asm("v_bitop3_b32 %0, %1, %2, %3 bitop3:0x96" : "=v"(n2_) : "v"((uint32_t)(p0 >> 32)), "v"(c3), "s"(k1));
        c1 = (uint32_t)p1; c3 = (uint32_t)p0; c0 = n0; c2 = n2_;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    o0 = c0; o1 = c1; o2 = c2; o3 = c3;
}

struct Stream {
    uint32_t k0, k1, c0, c1, c2, c3;   // c3 holds site/chain in the top 16 bits, block index below
    uint32_t b0, b1, b2, b3;
    int n;                               // unread words
    __device__ __forceinline__ Stream(uint64_t seed, uint32_t chain, uint32_t site, uint32_t i, uint32_t j, uint32_t sweep)
        : k0((uint32_t)seed), k1((uint32_t)(seed >> 32)), c0(i), c1(j), c2(sweep),
          c3((site << 24) | ((chain & 0xFFu) << 16)), b0(0), b1(0), b2(0), b3(0), n(0) {}
    __device__ __forceinline__ uint32_t next()
    {
        if (n == 0) {
            philox4x32_10(c0, c1, c2, c3, k0, k1, b0, b1, b2, b3);
            c3 = (c3 & 0xFFFF0000u) | ((c3 + 1u) & 0xFFFFu);
            n = 4;
        }
        const uint32_t v = b0;
        b0 = b1; b1 = b2; b2 = b3;
        --n;
        return v;
    }
};

template <typename real> __device__ __forceinline__ real uniform(Stream& s);
template <> __device__ __forceinline__ double uniform<double>(Stream& s) { return ((double)s.next() + 0.5) * (1.0 / 4294967296.0); }
template <> __device__ __forceinline__ float uniform<float>(Stream& s) { return ((float)(s.next() >> 9) + 0.5f) * (1.0f / 8388608.0f); }

template <typename real> __device__ __forceinline__ real expo(Stream& s) { return -q_log(uniform<real>(s)); }       // uniforms are normal numbers in (0, 1)

template <typename real> __device__ __forceinline__ real normal(Stream& s)
{
    const real u1 = uniform<real>(s), u2 = uniform<real>(s);
    return q_sqrt(real(-2) * q_log(u1)) * q_cos2pi(u2);
}

// fp64 cell path: the same variate from the range-specialised functions (uniforms are normal, finite numbers in (0, 1))
__device__ __forceinline__ double normal_sq_fast(Stream& s, const double2* tab = nullptr)      // N(0,1)^2 = -2 log(u1) cos^2(2 pi u2): the IG draw needs only the square
{
    const double u1 = uniform<double>(s), u2 = uniform<double>(s);
    const double c = fm::cos2pi(u2);
    return -2.0 * (tab ? fm::log(u1, tab) : fm::log(u1)) * (c * c);
}

// IG(mu, lambda), Michael-Schucany-Haas.  With y = N^2 and w = mu y the smaller root mu + mu/(2 lambda)(w - sqrt(w (4 lambda + w)))
// is evaluated as 4 lambda / (y (1 + sqrt(1 + 4 lambda / w))^2): no cancellation, no overflow, and the correct limits
// x1 -> lambda / y (Levy) as mu -> inf and x1 -> mu as y -> 0.  The first root is kept with probability mu/(mu + x1) = 1/(1 + x1/mu).
template <typename real> __device__ __forceinline__ real invgauss(Stream& s, real mu, real lambda)
{
    const real nrm = normal<real>(s);
    const real y = nrm * nrm;
    const real w = mu * y;
    const real t = real(1) + r_sqrt(real(1) + r_div(real(4) * lambda, w));
    const real x1 = r_div(real(4) * lambda, y * t * t);
    const real u = uniform<real>(s);
    return (u >= r_rcp(real(1) + x1 * r_rcp(mu))) ? r_div(mu * mu, x1) : x1;
}

// fp64 overload: y = N^2 without the square root, log and cosine from namespace fm; the divisions and the inner square root keep IEEE semantics
// (mu = inf, y -> 0 and w -> 0 must give the documented limits)
__device__ __forceinline__ double invgauss(Stream& s, double mu, double lambda, const double2* tab = nullptr)
{
    const double y = normal_sq_fast(s, tab);
    const double w = mu * y;
    const double t = 1.0 + sqrt(1.0 + 4.0 * lambda / w);
    const double x1 = 4.0 * lambda / (y * t * t);
    const double u = uniform<double>(s);
    return (u >= 1.0 / (1.0 + x1 / mu)) ? mu * mu / x1 : x1;
}

// nu = clamp(1 / IG(clamp(parB/parA, 1e-10, Inf), parB^2), 1e-10, 1e10): src/Draw.pl.jl:310-318, 333-341
template <typename real> __device__ __forceinline__ real qr_weight(Stream& s, real parA, real parB, [[maybe_unused]] const double2* tab = nullptr)
{
    real mu = r_div(parB, parA);
    mu = mu < real(1e-10) ? real(1e-10) : mu;
    real nu;
    if constexpr (sizeof(real) == 8) nu = r_rcp(invgauss(s, mu, parB * parB, tab)); else nu = r_rcp(invgauss(s, mu, real(parB * parB)));
    nu = nu < real(1e-10) ? real(1e-10) : (nu > real(1e10) ? real(1e10) : nu);
    return nu;
}

// The same weight on the fp64 engine's cell path, from range-specialised division / root / reciprocal (the IEEE sequences of the generic form are
// five divisions and a root: ~170 of its ~290 instructions).  absres = |residual|, cB = sqrt(2 k2 + k1^2), lambda = parB^2: parB / parA = cB / absres
// whatever the scale sqrt(sig2t k2) both were divided by.  An infinite mean (an exactly zero residual) is replaced by 1e150: the same root, the same
// branch (x1 -> lambda / y, second root never taken); the draw is clamped into the normal range before its reciprocal.
__device__ __forceinline__ double invgauss_q(Stream& s, double mu, double lambda, const double2* tab)       // mu in [1e-10, 1e150]
{
    const double y = normal_sq_fast(s, tab);
    const double w = mu * y;
    const double t = 1.0 + fm::sqrt(1.0 + fm::div(4.0 * lambda, w));
    const double x1 = fm::div(4.0 * lambda, y * t * t);
    const double u = uniform<double>(s);
    return (u >= fm::rcp(1.0 + x1 * fm::rcp(mu))) ? fm::div(mu * mu, x1) : x1;
}
__device__ __forceinline__ double qr_weight_q(Stream& s, double absres, double cB, double lambda, const double2* tab)
{
    double mu = absres > 1e-140 ? fm::div(cB, absres) : 1e150;
    mu = mu < 1e-10 ? 1e-10 : (mu > 1e150 ? 1e150 : mu);
    double x = invgauss_q(s, mu, lambda, tab);
    x = x < 1e-300 ? 1e-300 : (x > 1e300 ? 1e300 : x);
    const double nu = fm::rcp(x);
    return nu < 1e-10 ? 1e-10 : (nu > 1e10 ? 1e10 : nu);
}

// ---- standard normal quantile -----------------------------------------------------------------
// fp32: Giles' (2010) single-precision erfinv polynomial (relative error ~1.3e-7), argument formed from p without cancellation.
__device__ __forceinline__ float giles_erfinv_poly(float w)
{
    float pl;
    if (w < 5.0f) {
        w -= 2.5f;
        pl = 2.81022636e-08f; pl = fmaf(pl, w, 3.43273939e-07f); pl = fmaf(pl, w, -3.5233877e-06f); pl = fmaf(pl, w, -4.39150654e-06f);
        pl = fmaf(pl, w, 0.00021858087f); pl = fmaf(pl, w, -0.00125372503f); pl = fmaf(pl, w, -0.00417768164f); pl = fmaf(pl, w, 0.246640727f);
        pl = fmaf(pl, w, 1.50140941f);
    } else {
        w = __builtin_amdgcn_sqrtf(w) - 3.0f;
        pl = -0.000200214257f; pl = fmaf(pl, w, 0.000100950558f); pl = fmaf(pl, w, 0.00134934322f); pl = fmaf(pl, w, -0.00367342844f);
        pl = fmaf(pl, w, 0.00573950773f); pl = fmaf(pl, w, -0.0076224613f); pl = fmaf(pl, w, 0.00943887047f); pl = fmaf(pl, w, 1.00167406f);
        pl = fmaf(pl, w, 2.83297682f);
    }
    return pl;
}
__device__ __forceinline__ float ndtri(float p)
{
    const float x = 2.0f * p - 1.0f;
    const float w = -r_log(4.0f * p * (1.0f - p));
    return 1.41421356237f * giles_erfinv_poly(w) * x;
}
// fp64: Wichura's (1988) algorithm AS 241, routine PPND16 -- rational minimax approximations of degree 7/7 on three ranges, relative
// error < 1e-16 -- evaluated directly: no iteration, one logarithm and one square root outside the central range.  (The oracle reaches
// the same function by Newton steps on erfc, oracle/orc_rng.h::orc_ndtri; the two agree to a few ulp, which the parity tests check.)
namespace as241 {
// each range in two forms: numerator and denominator separately (the cell path folds the division into one it needs anyway), and their quotient
__device__ __forceinline__ void central(double r, double& n, double& d)      // |p - 1/2| <= 0.425, r = 0.180625 - (p - 1/2)^2; n / d = Phi^-1(p) / (p - 1/2)
{
    n = 2.5090809287301226727e+3; d = 5.2264952788528545610e+3;
    n = fma(n, r, 3.3430575583588128105e+4); d = fma(d, r, 2.8729085735721942674e+4);
    n = fma(n, r, 6.7265770927008700853e+4); d = fma(d, r, 3.9307895800092710610e+4);
    n = fma(n, r, 4.5921953931549871457e+4); d = fma(d, r, 2.1213794301586595867e+4);
    n = fma(n, r, 1.3731693765509461125e+4); d = fma(d, r, 5.3941960214247511077e+3);
    n = fma(n, r, 1.9715909503065514427e+3); d = fma(d, r, 6.8718700749205790830e+2);
    n = fma(n, r, 1.3314166789178437745e+2); d = fma(d, r, 4.2313330701600911252e+1);
    n = fma(n, r, 3.3871328727963666080e+0); d = fma(d, r, 1.0);
}
__device__ __forceinline__ void mid(double r, double& n, double& d)          // r = sqrt(-log(min(p, 1 - p))) <= 5; n / d = |Phi^-1(p)|
{
    r -= 1.6;
    n = 7.74545014278341407640e-4; d = 1.05075007164441684324e-9;
    n = fma(n, r, 2.27238449892691845833e-2); d = fma(d, r, 5.47593808499534494600e-4);
    n = fma(n, r, 2.41780725177450611770e-1); d = fma(d, r, 1.51986665636164571966e-2);
    n = fma(n, r, 1.27045825245236838258e+0); d = fma(d, r, 1.48103976427480074590e-1);
    n = fma(n, r, 3.64784832476320460504e+0); d = fma(d, r, 6.89767334985100004550e-1);
    n = fma(n, r, 5.76949722146069140550e+0); d = fma(d, r, 1.67638483018380384940e+0);
    n = fma(n, r, 4.63033784615654529590e+0); d = fma(d, r, 2.05319162663775882187e+0);
    n = fma(n, r, 1.42343711074968357734e+0); d = fma(d, r, 1.0);
}
__device__ __forceinline__ void far(double r, double& n, double& d)          // r > 5
{
    r -= 5.0;
    n = 2.01033439929228813265e-7; d = 2.04426310338993978564e-15;
    n = fma(n, r, 2.71155556874348757815e-5); d = fma(d, r, 1.42151175831644588870e-7);
    n = fma(n, r, 1.24266094738807843860e-3); d = fma(d, r, 1.84631831751005468180e-5);
    n = fma(n, r, 2.65321895265761230930e-2); d = fma(d, r, 7.86869131145613259100e-4);
    n = fma(n, r, 2.96560571828504891230e-1); d = fma(d, r, 1.48753612908506148525e-2);
    n = fma(n, r, 1.78482653991729133580e+0); d = fma(d, r, 1.36929880922735805310e-1);
    n = fma(n, r, 5.46378491116411436990e+0); d = fma(d, r, 5.99832206555887937690e-1);
    n = fma(n, r, 6.65790464350110377720e+0); d = fma(d, r, 1.0);
}
__device__ __forceinline__ double central(double r) { double n, d; central(r, n, d); return n / d; }
__device__ __forceinline__ double mid(double r) { double n, d; mid(r, n, d); return n / d; }
__device__ __forceinline__ double far(double r) { double n, d; far(r, n, d); return n / d; }
}  // namespace as241
__device__ __forceinline__ double ndtri(double p)
{
    const double q = p - 0.5;
    if (fabs(q) <= 0.425) return q * as241::central(0.180625 - q * q);
    const double r = sqrt(-log(q < 0.0 ? p : 1.0 - p));
    const double v = r <= 5.0 ? as241::mid(r) : as241::far(r);
    return q < 0.0 ? -v : v;
}

template <typename real> __device__ __forceinline__ real word_to_unif(uint32_t w);
template <> __device__ __forceinline__ double word_to_unif<double>(uint32_t w) { return ((double)w + 0.5) * (1.0 / 4294967296.0); }
// ((w >> 9) + 1/2) 2^-23, formed exactly without an int->float conversion: 1.m - (1 - 2^-24), both operands and the result representable
template <> __device__ __forceinline__ float word_to_unif<float>(uint32_t w) { return __uint_as_float((w >> 9) | 0x3F800000u) - 0.99999994f; }

// ---- Polya-Gamma PG(1, c) -------------------------------------------------------------------
// Single-level rejection sampler for J*(1, z): one attempt = one Philox block (u0..u3), no inner loop.  Envelope pieces:
//   x > t          : (pi/2) e^{-K x}, K = pi^2/8 + z^2/2, mass p = pi/(2K) e^{-K t};  X = t + E/K,  E = -log u1
//   x <= t, z < 8  : in Z = x^{-1/2} >= a = 1/sqrt(t) the piece is 4 phi(Z) e^{-z^2/(2 Z^2)}; Robert's exponential proposal for a normal tail,
//                    tilted per z:  Z = a + E/lam, kept with probability e^{g(Z) - M}, g(Z) = -(Z - lam)^2/2 - z^2/(2 Z^2), M >= max g;
//                    envelope mass q = 4/(lam sqrt(2 pi)) e^{lam^2/2 - lam a + M}.  (lam, 1/lam, M, q) come from a table over z-bins of width
//                    1/16 (pg_bin: Z*^2 = sqrt(10.6 + 1.07 z_k^2) for the bin's lower edge z_k, lam = Z* - z_k^2/Z*^3, M = g(Z*; z_k)).
//   x <= t, z >= 8 : the IG(1/z,1) kernel on all x > 0, mass 2 e^{-z}; draws with x > t rejected (|eta| >= 16: the reference form only)
// followed by the alternating-series test in ratio form (rho_n = a_n/a_0).  Specification shared with oracle/orc_rng.h.
// ONE logarithm of u1 serves both pieces of the common case and the value of an accepted draw is that logarithm and one reciprocal,
// whatever the piece: no normal quantile, no piece-specific fp64 evaluation, nothing to sort by piece.  (Round 2 proposed the left
// piece by the inverse normal cdf: 0.999 instead of 0.95 acceptance at z = 0, but three quantile ranges whose fp64 values had to be queued
// by piece -- 648 VALU instructions per cell-update against DESIGN.md 4b's figure for this form.)
constexpr double PG_ZMAX = 8.0;

// the table entry of z-bin k: {lam, c = 1/lam, M, q}.  Host and device (the engine uploads the host's copy: one table per engine)
__host__ __device__ inline void pg_bin(int k, double* out4)
{
    const double zk = (double)k / 16.0;
    const double Zs2 = ::sqrt(10.6 + 1.07 * zk * zk), Zs = ::sqrt(Zs2);
    const double d = zk * zk / (Zs * Zs2);
    const double l = Zs - d, m = -0.5 * d * d - 0.5 * zk * zk / Zs2;
    out4[0] = l; out4[1] = 1.0 / l; out4[2] = m;
    out4[3] = 4.0 / (l * ::sqrt(2.0 * 3.14159265358979323846)) * ::exp(0.5 * l * l - 1.25 * l + m);
}
__device__ __forceinline__ int pg_bin_index(double z) { return (int)fmin(z * 16.0, (double)(PG_NBIN - 1)); }       // NaN -> the last bin (unused: z >= 8 takes the reference form)
__device__ __forceinline__ int pg_bin_index(float z) { return (int)fminf(z * 16.0f, (float)(PG_NBIN - 1)); }

// probability that an attempt proposes from the exponential tail (debug sampler 7); gtab: the [PG_NBIN][4] table in global memory
__device__ __forceinline__ double pg_tail_weight(double z, const double* gtab)
{
    const double PId = 3.14159265358979323846;
    const double K = 0.125 * PId * PId + 0.5 * z * z;
    const double p = PId / (2.0 * K) * exp(-K * 0.64);
    const double q = z < PG_ZMAX ? gtab[4 * pg_bin_index(z) + 3] : 2.0 * exp(-z);
    return p / (p + q);
}

// Reference form of one attempt: the specification, statement by statement, in fp64 throughout.  The row pass reaches it for the few
// attempts in 10^5 that fall inside a guard band of the fast forms below and for z >= 8.
__device__ __forceinline__ bool pg1_attempt_ref(double z, uint32_t w0, uint32_t w1, uint32_t w2, uint32_t w3, const double* gtab, double& out)
{
    const double t = 0.64, PI = 3.14159265358979323846;
    const double K = 0.125 * PI * PI + 0.5 * z * z;
    const double u0 = word_to_unif<double>(w0), u1 = word_to_unif<double>(w1), u2 = word_to_unif<double>(w2), V = word_to_unif<double>(w3);
    const double p = PI / (2.0 * K) * exp(-K * t);
    double x;
    bool ok = true;
    if (z < PG_ZMAX) {
        const double* b = gtab + 4 * pg_bin_index(z);
        const double lam = b[0], c = b[1], M = b[2], q = b[3];
        if (u0 < p / (p + q)) x = t + (-log(u1)) / K;
        else {
            const double Z = 1.25 + c * (-log(u1));
            x = 1.0 / (Z * Z);
            ok = !(u2 > exp(-0.5 * (Z - lam) * (Z - lam) - 0.5 * z * z * x - M));
        }
    } else if (u0 < p / (p + 2.0 * exp(-z))) {
        x = t + (-log(u1)) / K;
    } else {
        const double mu = 1.0 / z, nrm = ndtri(u1);
        const double ww = mu * nrm * nrm;
        const double sq = sqrt(ww) * sqrt(4.0 + ww), den = sq + ww;
        const double q = den > 0.0 ? 2.0 * sqrt(ww) / den : 1.0;
        const double x1 = mu * q * q;
        x = (u2 >= mu / (mu + x1)) ? mu * mu / x1 : x1;
        ok = !(x > t);
    }
    out = 0.25 * x;
    if (!ok) return false;
    // alternating series: accept at odd n if V <= S_n, reject at even n if V > S_n
    const double e1 = (x > t) ? -0.5 * PI * PI * x : -2.0 / x;     // rho_n = (2n+1) exp(n(n+1) e1)
    double S = 1.0 - 3.0 * exp(2.0 * e1);
    if (V <= S) return true;
    for (int n = 2; n <= 200; ++n) {
        const double rho = (double)(2 * n + 1) * exp((double)(n * (n + 1)) * e1);
        if (n & 1) { S -= rho; if (V <= S) return true; }
        else       { S += rho; if (V > S) return false; }
    }
    return true;
}
// ... as a REAL call: inlined into the attempt loop its OCML exponentials and logarithms would set that loop's register pressure.  Behind a
// call the callee's registers are its own, and the caller saves its live values only on the rare path that calls.
__device__ __attribute__((noinline)) bool pg1_attempt_ref_call(double z, uint32_t w0, uint32_t w1, uint32_t w2, uint32_t w3, const double* gtab, double* out)
{
    double o;
    const bool a = pg1_attempt_ref(z, w0, w1, w2, w3, gtab, o);
    *out = o;
    return a;
}

// fp32 fast mode, z < 8: the same attempt as near-straight-line code for a 64-wide wave.  bin = {lam, 1/lam, M, q} of the cell's z-bin.
//   * the mixture test u0 < p/(p+q) is evaluated as u0 (p+q) < p (no reciprocal); 1/K is shared by p and the tail proposal;
//   * E = -log u1 serves the tail (X = t + E/K) and the left piece (Z = a + E/lam), whose series exponent -2/X = -2 Z^2 needs no reciprocal;
//   * the series stops after its first term: rho_2 = 5 e^{6 e1} <= 3.6e-8 for every x an attempt can propose (e^{2 e1} <= 1.93e-3), below
//     the resolution of fp32 at S_1 ~ 1, so V <= S_1 decides.
__device__ __forceinline__ bool pg1_attempt(float z, uint32_t w0, uint32_t w1, uint32_t w2, uint32_t w3, float4 bin, float& out)
{
    const float t = 0.64f, PI = 3.14159265358979f;
    const float K = fmaf(0.5f * z, z, 0.125f * PI * PI);
    const float rK = r_rcp(K);
    const float p = (0.5f * PI) * rK * r_exp(-K * t);
    const float u0 = word_to_unif<float>(w0), u1 = word_to_unif<float>(w1), u2 = word_to_unif<float>(w2), V = word_to_unif<float>(w3);
    const bool tail = u0 * (p + bin.w) < p;
    const float E = -r_log(u1);
    const float Z = fmaf(bin.y, E, 1.25f), Z2 = Z * Z;
    const float xl = r_rcp(Z2), dz = Z - bin.x;
    const float g = fmaf(-0.5f * dz, dz, fmaf(-0.5f * z * z, xl, -bin.z));       // g(Z) - M <= 0
    const bool okl = !(u2 > r_exp(g));
    const float xt = fmaf(E, rK, t);
    out = 0.25f * (tail ? xt : xl);
    const float e1 = tail ? -0.5f * PI * PI * xt : -2.0f * Z2;
    const float S = 1.0f - 3.0f * r_exp(2.0f * e1);
    return (tail || okl) && V <= S;
}

// fp64 engine, z < 8: the same attempt with its accept / reject DECISIONS taken in fp32 and its VALUE in fp64.  The three comparisons of an
// attempt (tail or left piece; left proposal kept; first term of the alternating series) are evaluated with the hardware-rate fp32
// instructions; each comes with a guard band several times wider than the worst fp32 error of its two sides (derivations in DESIGN.md 4b):
// outside the band the fp32 outcome IS the fp64 outcome; a lane inside a band (about 1 in 10^4 attempts) or with z >= 8 is flagged `unsure`
// and the caller repeats the attempt through pg1_attempt_ref.  The value of the proposal -- both pieces: the fp64 logarithm of u1 and one
// reciprocal -- is evaluated for every lane of the trip (nine lanes in ten accept), so there is nothing to queue or sort.
//   the series needs no second term: rho_2 <= 3.6e-8 is smaller than the band on V <= S_1, so V > S_1 + band implies V > S_2 (reject) and
//   V <= S_1 - band implies accept.
// bin = {lam, 1/lam, M, q} rounded to fp32, cd = 1/lam in fp64 (the value's Z = a + E/lam).  TAB: fm::log through the LDS table.
template <bool TAB>
__device__ __forceinline__ bool pg1_attempt_f64(double z, uint32_t w0, uint32_t w1, uint32_t w2, uint32_t w3, float4 bin, double cd,
                                                [[maybe_unused]] const double2* tab, double& out, bool& unsure)
{
    const float t = 0.64f, PI = 3.14159265358979f;
    const float zf = (float)z;
    const float K = fmaf(0.5f * zf, zf, 0.125f * PI * PI);
    const float rK = r_rcp(K);
    const float p = (0.5f * PI) * rK * r_exp(-K * t);
    const float u0 = word_to_unif<float>(w0), u2 = word_to_unif<float>(w2), V = word_to_unif<float>(w3);
    // E = -log u1 in fp64 first: the value needs it anyway, and rounded to fp32 it serves the decisions (no v_log_f32, and E is exact to fp32 rounding)
    double Ed;
    if constexpr (TAB) Ed = -fm::log_word(w1, tab); else Ed = -fm::log(word_to_unif<double>(w1));
    // (a) tail piece iff u0 (p + q) < p
    const float da = fmaf(u0, p + bin.w, -p);
    const bool tail = da < 0.0f;
    bool uns = !(fabsf(da) > fmaf(2e-5f, p, 5e-6f * bin.w)) || !(z < PG_ZMAX);
    // (b) left proposal kept iff u2 <= e^{g(Z) - M}
    const float E = (float)Ed;
    const float Z = fmaf(bin.y, E, 1.25f), Z2 = Z * Z;
    const float xl = r_rcp(Z2), dz = Z - bin.x;
    const float thr = r_exp(fmaf(-0.5f * dz, dz, fmaf(-0.5f * zf * zf, xl, -bin.z)));
    uns = uns || (!tail && fabsf(u2 - thr) <= 3e-5f);
    const bool ok = tail || !(u2 > thr);
    // (c) V <= S_1 = 1 - 3 e^{2 e1}
    const float xt = fmaf(E, rK, t);
    const float e1 = tail ? -0.5f * PI * PI * xt : -2.0f * Z2;
    const float S = 1.0f - 3.0f * r_exp(2.0f * e1);
    unsure = uns || (ok && fabsf(V - S) <= 1e-6f);
    // the value: tail X = t + E/K, left X = 1/Z^2 with Z = a + E/lam -- E/den or 1/den from ONE reciprocal
    const double PId = 3.14159265358979323846;
    const double Zd = fma(cd, Ed, 1.25);
    const double den = tail ? fma(0.5 * z, z, 0.125 * PId * PId) : Zd * Zd;
    const double r = fm::rcp(den);
    const double qd = Ed * r;
    out = 0.25 * (tail ? 0.64 + fma(fma(-den, qd, Ed), r, qd) : r);
    return ok && V <= S;
}

// one whole attempt in the calling lane's own control flow (debug sampler); gtab: the [PG_NBIN][4] table in global memory
__device__ __forceinline__ bool pg1_attempt(double z, uint32_t w0, uint32_t w1, uint32_t w2, uint32_t w3, const double* gtab, double& out)
{
    const double* b = gtab + 4 * pg_bin_index(z);
    bool unsure;
    bool accept = pg1_attempt_f64<false>(z, w0, w1, w2, w3, make_float4((float)b[0], (float)b[1], (float)b[2], (float)b[3]), b[1], nullptr, out, unsure);
    if (unsure) accept = pg1_attempt_ref(z, w0, w1, w2, w3, gtab, out);
    return accept;
}
__device__ __forceinline__ bool pg1_attempt(float z, uint32_t w0, uint32_t w1, uint32_t w2, uint32_t w3, const double* gtab, float& out)
{
    if (!(z < (float)PG_ZMAX)) { double o; const bool a = pg1_attempt_ref((double)z, w0, w1, w2, w3, gtab, o); out = (float)o; return a; }
    const double* b = gtab + 4 * pg_bin_index(z);
    return pg1_attempt(z, w0, w1, w2, w3, make_float4((float)b[0], (float)b[1], (float)b[2], (float)b[3]), out);
}

// draw addressed by a stream: attempt k consumes block k
template <typename real> __device__ __forceinline__ real pg1(Stream& s, real c, const double* gtab)
{
    const real z = real(0.5) * r_abs(c);
    real out = real(0);
    for (int tries = 0; tries < MAX_TRIES; ++tries) {
        const uint32_t w0 = s.next(), w1 = s.next(), w2 = s.next(), w3 = s.next();
        if (pg1_attempt(z, w0, w1, w2, w3, gtab, out)) break;
    }
    return out;
}

// ---- item-level samplers (fp64 only) --------------------------------------------------------
__device__ __forceinline__ double truncnorm0(Stream& s, double m, double sd)
{
    // (the item-level draws are dependent fp64 chains on a handful of lanes -- the head of every sweep waits for them: range-specialised
    // division / root / exponential instead of the IEEE and OCML sequences, cf. namespace fm)
    const double alpha = -q_div(m, sd);
    double z;
    int tries = 0;
    if (alpha <= 0.0) {
        do { z = normal<double>(s); } while (z < alpha && ++tries < MAX_TRIES);
    } else {
        const double lam = 0.5 * (alpha + q_sqrt(alpha * alpha + 4.0));
        const double ilam = q_rcp(lam);
        for (;;) {
            z = alpha + expo<double>(s) * ilam;
            const double u = uniform<double>(s);
            if (u <= q_exp_neg(0.5 * (z - lam) * (z - lam)) || ++tries >= MAX_TRIES) break;
        }
    }
    return m + sd * z;
}

__device__ __forceinline__ double gamma_mt(Stream& s, double shape)
{
    const double d = shape - 1.0 / 3.0, c = q_rcp(q_sqrt(9.0 * d));
    for (int tries = 0;; ++tries) {
        double x, v;
        do { x = normal<double>(s); v = 1.0 + c * x; } while (v <= 0.0 && ++tries < MAX_TRIES);
        v = v * v * v;
        const double u = uniform<double>(s);
        const double x2 = x * x;
        if (u < 1.0 - 0.0331 * x2 * x2) return d * v;         // Marsaglia-Tsang squeeze: implies the log test below, so it changes no decision
        if (!(v >= 1e-300)) { if (tries >= MAX_TRIES) return __builtin_nan(""); continue; }      // (1 + c x)^3 underflowed: the log test below fails for every u
        if (q_log(u) < 0.5 * x2 + d - d * v + d * q_log(v)) return d * v;
        if (tries >= MAX_TRIES) return __builtin_nan("");
    }
}
__device__ __forceinline__ double invgamma(Stream& s, double shape, double scale) { return q_div(scale, gamma_mt(s, shape)); }
__device__ __forceinline__ double chisq(Stream& s, double k) { return 2.0 * gamma_mt(s, 0.5 * k); }

// GIG(p, a, b), density proportional to x^(p-1) exp(-(a x + b/x)/2): the distribution type of src/GenInvGaussian.jl (dead code in the
// reference), by Devroye's (2014) rejection sampler for the log-concave density of log X.  Same specification as
// oracle/orc_rng.h::orc_gig; one attempt uses three uniforms.
__device__ __forceinline__ double gig_psi(double x, double alpha, double lam) { return -alpha * (cosh(x) - 1.0) - lam * (exp(x) - x - 1.0); }
__device__ __forceinline__ double gig_dpsi(double x, double alpha, double lam) { return -alpha * sinh(x) - lam * (exp(x) - 1.0); }
__device__ inline double gig(Stream& st, double p, double a, double b)
{
    const double omega = sqrt(a * b);
    const bool inv = p < 0.0;
    const double lam = fabs(p);
    const double alpha = sqrt(omega * omega + lam * lam) - lam;
    double x = -gig_psi(1.0, alpha, lam), t, s;
    if (x >= 0.5 && x <= 2.0) t = 1.0; else if (x > 2.0) t = sqrt(2.0 / (alpha + lam)); else t = log(4.0 / (alpha + 2.0 * lam));
    x = -gig_psi(-1.0, alpha, lam);
    if (x >= 0.5 && x <= 2.0) s = 1.0;
    else if (x > 2.0) s = sqrt(4.0 / (alpha * cosh(1.0) + lam));
    else { const double s1 = 1.0 / lam, s2 = log(1.0 + 1.0 / alpha + sqrt(1.0 / (alpha * alpha) + 2.0 / alpha)); s = s1 < s2 ? s1 : s2; }
    const double eta = -gig_psi(t, alpha, lam), zeta = -gig_dpsi(t, alpha, lam);
    const double theta = -gig_psi(-s, alpha, lam), xi = gig_dpsi(-s, alpha, lam);
    const double pp = 1.0 / xi, r = 1.0 / zeta, td = t - r * eta, sd = s - pp * theta, q = td + sd;
    double rnd = 0.0;
    for (int tries = 0; tries < MAX_TRIES; ++tries) {
        const double U = uniform<double>(st), V = uniform<double>(st), W = uniform<double>(st);
        if (U < q / (pp + q + r)) rnd = -sd + q * V;
        else if (U < (q + r) / (pp + q + r)) rnd = td - r * log(V);
        else rnd = -sd + pp * log(V);
        const double f1 = exp(-eta - zeta * (rnd - t)), f2 = exp(-theta + xi * (rnd + s));
        const double g = (rnd >= -sd && rnd <= td) ? 1.0 : (rnd > td ? f1 : f2);
        if (W * g <= exp(gig_psi(rnd, alpha, lam))) break;
    }
    double y = exp(rnd) * (lam / omega + sqrt(1.0 + lam * lam / (omega * omega)));
    if (inv) y = 1.0 / y;
    return y * sqrt(b / a);
}

}  // namespace erm
