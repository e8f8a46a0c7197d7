// Device-side building blocks shared by the gfx950 kernels: IEEE division by a per-member
// constant with the reciprocal hoisted out of the RK4 loops, and the classical RK4 update in
// the reference's exact association.
//
// Compiled with -ffp-contract=off: the EXACT paths must round like rustc's output (which never
// fuses a*b+c); FMAs appear only where written as __builtin_fma.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace rscm {

// ---------------------------------------------------------------------------------------------
// n / d for a d that is constant over thousands of divisions.
//
// hipcc lowers an f64 division to
//     sd = div_scale(d), sn = div_scale(n)
//     r  = rcp(sd); e = fma(-sd,r,1); r = fma(r,e,r); e = fma(-sd,r,1); r = fma(r,e,r)
//     q  = sn*r; rem = fma(-sd,q,sn); res = div_fmas(rem,r,q); div_fixup(res,d,n)
// (11 VALU instructions, 58 cycles per wavefront measured; tools/valu_microbench.hip).
// v_div_scale_f64 leaves both operands unscaled and v_div_fixup_f64 passes the quotient through
// when the divisor is normal with a normal reciprocal, exponent(n) - exponent(d) < 768, the
// quotient is not denormal and the numerator's biased exponent exceeds 53.  All of that holds
// when
//     biased exponent of d in [895, 1151]   (|d| in [2^-128, 2^129))      "divisor window"
//     biased exponent of n in [512, 1535]   (|n| in [2^-511, 2^513))      "numerator window"
// and then the compiler's sequence is exactly
//     q = n*r; rem = fma(-d,q,n); res = fma(rem,r,q)
// with r depending on d alone.  So r is computed once per member (same instruction sequence)
// and each division costs three instructions.  Numerators outside the window (zeros,
// denormals, values on their way to overflow, inf, NaN) must take the compiler's division;
// kernels either branch per division (div_const) or validate a whole model year at once
// (two_layer.hip).  tests/test_gpu_parity.py checks the identity on random and edge-case
// operands through rscm_gpu_selftest_div.
// ---------------------------------------------------------------------------------------------

// tag < 0  <=>  biased exponent of x in [512, 1535]: (hi << 1) moves exponent bit 10 to bit 31
// and bit 9 to bit 30; adding 2^30 sets bit 31 exactly for the bit pairs 01 and 10.
__device__ __forceinline__ int32_t window_tag(double x)
{
    // One v_lshl_add_u32 on the high dword (0x40000000 is the inline constant 2.0).  Written as
    // asm because LLVM canonicalises ((hi << 1) + c) into alignbit + and + add (3 instructions).
    int32_t t;
    asm("v_lshl_add_u32 %0, %1, 1, 2.0" : "=v"(t) : "v"(__double2hiint(x)));
    return t;
}

__device__ __forceinline__ bool divisor_in_window(double d)
{
    const uint32_t ed = ((uint32_t)__double2hiint(d) >> 20) & 0x7FFu;
    return (ed - 895u) <= 256u;
}

__device__ __forceinline__ double refined_rcp(double d)
{
    double r = __builtin_amdgcn_rcp(d);
    double e = __builtin_fma(-d, r, 1.0);
    r = __builtin_fma(r, e, r);
    e = __builtin_fma(-d, r, 1.0);
    r = __builtin_fma(r, e, r);
    return r;
}

// 1/d for any d: the refined reciprocal inside the divisor window, IEEE division outside it
// (rcp(+-inf) = 0 and rcp(0) = inf would turn the refinement's fma(-d, r, 1) into NaN, where
// the reference's x / d gives 0 or inf).
__device__ __forceinline__ double guarded_rcp(double d)
{
    double r = refined_rcp(d);
    if (__builtin_expect(!divisor_in_window(d), 0)) r = 1.0 / d;
    return r;
}

// The three-instruction quotient; equals n/d bit for bit inside the two windows.
__device__ __forceinline__ double spec_div(double n, double d, double r)
{
    const double q = n * r;
    return __builtin_fma(__builtin_fma(-d, q, n), r, q);
}

struct ConstDiv {
    double d;  // divisor
    double r;  // refined reciprocal
    bool ok;   // divisor inside its window
};

__device__ __forceinline__ ConstDiv make_const_div(double d)
{
    ConstDiv c;
    c.d = d;
    c.r = refined_rcp(d);
    c.ok = divisor_in_window(d);
    return c;
}

__device__ __forceinline__ bool const_div_fast_ok(double n, const ConstDiv& c)
{
    return c.ok && window_tag(n) < 0;
}

// Branch-per-division form (used where a year cannot cheaply be replayed).
__device__ __forceinline__ double div_const(double n, const ConstDiv& c)
{
    double res = spec_div(n, c.d, c.r);
    if (__builtin_expect(!const_div_fast_ok(n, c), 0)) res = n / c.d;
    return res;
}

// ---------------------------------------------------------------------------------------------
// Classical RK4 combination, ode_solvers 0.6.1 association (crates/rscm-core/src/ivp/mod.rs:245-253
// hands the system to Rk4::new(f, t0, y0, t1, h)):
//     y' = y + (((k1 + k2*2) + k3*2) + k4) * (h/6)
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ double rk4_combine(double y, double k1, double k2, double k3, double k4,
                                              double sixth)
{
    return y + (((k1 + k2 * 2.0) + k3 * 2.0) + k4) * sixth;
}

// Same value while 2*k2 and 2*k3 cannot overflow (doubling is exact, so fusing it into the
// following addition does not change the rounding).
__device__ __forceinline__ double rk4_combine_fused2(double y, double k1, double k2, double k3,
                                                     double k4, double sixth)
{
    return y + (__builtin_fma(k3, 2.0, __builtin_fma(k2, 2.0, k1)) + k4) * sixth;
}

// ---------------------------------------------------------------------------------------------
// ln x for the forcing formulas (CO2ERF in the coupled chain and as a linked component: once per member and model step).
// The device library's log is 75 instructions, 46 of them separately rounded additions of its double-double arithmetic; this is
// the classical reduction  x = 2^k m,  m in [sqrt(1/2), sqrt(2)),  f = m - 1,  s = f / (2 + f),
//     ln m = f - f^2/2 + s (f^2/2 + R(s^2)),   R an odd minimax polynomial (degree 14 in s; W. Kahan's coefficients as published in
// the freely distributable fdlibm e_log.c, error < 1 ulp), with the quotient as reciprocal estimate + one Newton step + one
// correction of the quotient: ~40 instructions.  Same accuracy class as the library's (<= 1 ulp), not the same bits: every use sits in a
// tolerance-parity kind (tests/test_gpu_parity.py: 1e-11).  Zero, negatives, denormals, inf and NaN take the library's log.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ double log_f64(double x)
{
    if (__builtin_expect(!(x >= 2.2250738585072014e-308 && x < __builtin_inf()), 0)) return log(x);
    double m = __builtin_amdgcn_frexp_mant(x);          // [0.5, 1)
    int32_t k = __builtin_amdgcn_frexp_exp(x);
    const bool low = m < 0.70710678118654752440;
    m = low ? m + m : m;                                 // [sqrt(1/2), sqrt(2))
    k = low ? k - 1 : k;
    const double dk = (double)k;
    const double f = m - 1.0;
    const double d = 2.0 + f;
    double r = __builtin_amdgcn_rcp(d);
    r = __builtin_fma(__builtin_fma(-d, r, 1.0), r, r);
    const double q = f * r;
    const double s = __builtin_fma(__builtin_fma(-d, q, f), r, q);
    const double z = s * s;
    const double w = z * z;
    const double t1 = w * __builtin_fma(w, __builtin_fma(w, 1.531383769920937332e-01, 2.222219843214978396e-01), 3.999999999940941908e-01);
    const double t2 = z * __builtin_fma(w, __builtin_fma(w, __builtin_fma(w, 1.479819860511658591e-01, 1.818357216161805012e-01),
                                                         2.857142874366239149e-01), 6.666666666666735130e-01);
    const double R = t2 + t1;
    const double hfsq = 0.5 * f * f;
    return dk * 6.93147180369123816490e-01 - ((hfsq - (s * (hfsq + R) + dk * 1.90821492927058770002e-10)) - f);
}

__device__ __forceinline__ int32_t max3_i32(int32_t a, int32_t b, int32_t c)
{
    return max(max(a, b), c);  // v_max3_i32
}

__device__ __forceinline__ bool is_finite(double x)
{
    return __builtin_isfinite(x);
}

}  // namespace rscm
