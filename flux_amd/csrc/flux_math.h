// flux_math.h -- FP64 elementary functions for the FAST render path on gfx950.
//
// The render loop is FP64-VALU bound (DESIGN.md "Kernels"), and most of its instructions sit in
// IEEE division (v_div_scale/v_rcp/6 fma/v_div_fmas/v_div_fixup), sqrt, and OCML's correctly-rounded-ish
// pow (~220 VALU) / sincos (~150).  These replacements keep full double precision (<= ~2 ulp, measured
// on the device by tests/test_gpu_fastmath.py through flux_debug_fastmath) but drop the scaling /
// special-case scaffolding the render loop's operand ranges never need:
//   frsqrt   v_rsq_f64 seed + one cubic (Halley) step                      6 VALU  (sqrt+3 div ~ 50)
//   fsqrt    v_rsq_f64 seed + Goldschmidt step + residual correction       9 VALU  (OCML 22)
//   fdiv     v_rcp_f64 seed + cubic step + residual correction             8 VALU  (IEEE 11)
//   flog2    frexp + s=(m-1)/(m+1) + degree-7 polynomial in s^2           ~28 VALU  (OCML 89)
//   fexp2    rndne + degree-11 polynomial + ldexp                         ~16 VALU  (OCML 42)
//   fpow     fexp2(y*flog2(x)), x >= 0                                    ~48 VALU  (OCML 224)
//   fsincos2pi  exact quarter-turn reduction of 2*pi*x + degree-6/7 polynomials  ~32 VALU (OCML 154)
// Polynomial coefficients: flux_math_coeffs.h (generated, scripts/fit_fast_math.py).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "flux_math_coeffs.h"

namespace flux {
namespace fastmath {

__device__ __forceinline__ double ffma(double a, double b, double c) { return __builtin_fma(a, b, c); }

// 1/sqrt(x), x > 0 finite normal
__device__ __forceinline__ double frsqrt(double x) {
    const double y0 = __builtin_amdgcn_rsq(x);
    const double t = x * y0;
    const double e = ffma(-t, y0, 1.0);              // 1 - x*y0^2
    const double a = ffma(0.375, e, 0.5);            // y0*(1 + e/2 + 3e^2/8): error O(e^3)
    return ffma(y0 * e, a, y0);
}

// sqrt(x), x >= 0; sqrt(0) = 0 exactly.  The tiny NEGATIVES that 1 - c*c or a discriminant can round to are taken as 0:
// unclamped, a negative x would run the iteration with g = x*y0 hugely negative and overflow to NaN, and one NaN sample
// poisons a pixel's whole sum.  (The seed is taken of max(x, denorm_min) so that it stays finite: 0 * rsq(denorm_min) = 0.)
__device__ __forceinline__ double fsqrt(double x) {
    x = __builtin_fmax(x, 0.0);
    // the seed of x + denorm_min (an INLINE constant of the instruction: the integer 1): x itself for every normal x, finite for x = 0
    // (until round 6: max(x, 1e-300) -- two scalar moves for the literal and a second maximum)
    const double y0 = __builtin_amdgcn_rsq(x + 4.9406564584124654e-324);
    double g = x * y0;
    double h = 0.5 * y0;
    const double r = ffma(-h, g, 0.5);
    g = ffma(g, r, g);
    h = ffma(h, r, h);
    const double d = ffma(-g, g, x);
    return ffma(d, h, g);
}

// a/b for finite normal b (no scaling: |b| well inside the exponent range)
__device__ __forceinline__ double fdiv(double a, double b) {
    const double r0 = __builtin_amdgcn_rcp(b);
    const double e = ffma(-b, r0, 1.0);
    const double c = r0 * e;
    const double r = r0 + ffma(c, e, c);             // r0*(1 + e + e^2)
    const double q = a * r;
    const double rem = ffma(-q, b, a);
    return ffma(rem, r, q);
}

// sqrt(x) for a caller that KNOWS x >= 0 (a discriminant behind its own `dq >= 0` test): the same iteration without the clamp of
// negative inputs (and without the canonicalising v_max the compiler puts in front of it): two instructions less, the same bits.
__device__ __forceinline__ double fsqrt_nonneg(double x) {
    // the seed of x + denorm_min: x itself, exactly, for every normal x; finite for x = 0 (0 * rsq(denorm_min) = 0); one addition
    // where max(x, denorm_min) is two instructions (the compiler canonicalises a maximum's operand first)
    const double y0 = __builtin_amdgcn_rsq(x + 4.9406564584124654e-324);
    double g = x * y0;
    double h = 0.5 * y0;
    const double r = ffma(-h, g, 0.5);
    g = ffma(g, r, g);
    h = ffma(h, r, h);
    const double d = ffma(-g, g, x);
    return ffma(d, h, g);
}

template <int N>
__device__ __forceinline__ double horner(const double (&c)[N], double x) {
    double r = c[N - 1];
#pragma unroll
    for (int k = N - 2; k >= 0; --k) r = ffma(r, x, c[k]);
    return r;
}

// log2(x), x > 0 finite normal.  (x == 0 returns a large negative finite number, see fpow.)
__device__ __forceinline__ double flog2(double x) {
    double m = __builtin_amdgcn_frexp_mant(x);        // [0.5, 1)
    int k = __builtin_amdgcn_frexp_exp(x);
    const bool lo = m < 0.70710678118654752440;
    m = lo ? m + m : m;                                // [sqrt(.5), sqrt(2))
    k = lo ? k - 1 : k;
    const double s = fdiv(m - 1.0, m + 1.0);
    const double w = s * s;
    const double p = poly_log2(w);
    return ffma(s, p, (double)k);
}

// 2^t for t <= ~1000 (clamped below at -1100 -> 0)
__device__ __forceinline__ double fexp2(double t) {
    t = __builtin_fmax(t, -1100.0);
    const double n = __builtin_rint(t);
    const double f = t - n;
    const double p = poly_exp2(f);
    return __builtin_amdgcn_ldexp(p, (int)n);
}

// fexp2 with the polynomial's coefficients in memory (flux_math_coeffs.h poly_exp2_tab)
// (no clamp of t: the one caller's argument is log2(1 - y) / (e + 1) of a TABULATED sample -- flog2 returns a finite value for every input,
// zero included (see there), and |log2| of a double is below 1100, so the bound of fexp2 cannot bind; the instruction and its 64-bit
// literal went in round 6)
__device__ __forceinline__ double fexp2_tab(double t, const double *c) {
    const double n = __builtin_rint(t);
    const double f = t - n;
    const double p = poly_exp2_tab(f, c);
    return __builtin_amdgcn_ldexp(p, (int)n);
}

// pow(x, y) for x >= 0, y > 0 finite (the render loop's domain: x = 1 - sample.y or a cosine).
__device__ __forceinline__ double fpow_pos(double x, double y) {
    const double r = fexp2(y * flog2(x));
    return x > 0.0 ? r : 0.0;
}

// (sin, cos) of 2*pi*x, x in [0, 1]: quarter-turn reduction is exact in f64
__device__ __forceinline__ void fsincos2pi(double x, double &s, double &c) {
    const double n = __builtin_rint(4.0 * x);
    const double f = ffma(4.0, x, -n);               // exact, |f| <= 1/2
    const double w = f * f;
    const double s0 = f * poly_sinq(w);
    const double c0 = poly_cosq(w);
    const int q = (int)n;
    const bool swap = (q & 1) != 0;
    double ss = swap ? c0 : s0;
    double cc = swap ? s0 : c0;
    // quadrant signs: sin negative for q = 2,3; cos negative for q = 1,2
    ss = (q & 2) ? -ss : ss;
    cc = ((q + 1) & 2) ? -cc : cc;
    s = ss;
    c = cc;
}

}  // namespace fastmath
}  // namespace flux
