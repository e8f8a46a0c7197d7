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
// (11 VALU instructions, v_rcp_f64 at quarter rate).  When div_scale leaves both operands
// unscaled and div_fixup passes the quotient through -- true whenever the biased exponent of d
// is in [895,1151] and that of n in [256,1535] -- the result is exactly
//     q = n*r; rem = fma(-d,q,n); res = fma(rem,r,q)
// with r depending on d alone.  So r is computed once per member (same instruction sequence)
// and each division costs three instructions; numerators outside the window (zeros, denormals,
// huge values on the way to overflow, inf, NaN) take the compiler's full division, so results
// are bit-identical to IEEE division for every input.  tests/test_gpu_parity.py checks this
// identity on random and edge-case operands through rscm_gpu_selftest_div.
// ---------------------------------------------------------------------------------------------
struct ConstDiv {
    double d;       // divisor
    double r;       // refined reciprocal
    uint32_t span;  // width of the accepted numerator-exponent window (0: never use the fast path)
};

constexpr uint32_t kNumExpLo = 256u << 20;          // numerator biased exponent >= 256
constexpr uint32_t kNumExpSpan = (1536u - 256u) << 20;  // and < 1536

__device__ __forceinline__ double refined_rcp(double d)
{
    double r = __builtin_amdgcn_rcp(d);
    double e = __builtin_fma(-d, r, 1.0);
    r = __builtin_fma(r, e, r);
    e = __builtin_fma(-d, r, 1.0);
    r = __builtin_fma(r, e, r);
    return r;
}

__device__ __forceinline__ ConstDiv make_const_div(double d)
{
    ConstDiv c;
    c.d = d;
    c.r = refined_rcp(d);
    const uint32_t ed = ((uint32_t)__double2hiint(d) >> 20) & 0x7FFu;
    c.span = (ed - 895u) <= 256u ? kNumExpSpan : 0u;
    return c;
}

__device__ __forceinline__ bool const_div_fast_ok(double n, const ConstDiv& c)
{
    const uint32_t en = (uint32_t)__double2hiint(n) & 0x7FF00000u;
    return (en - kNumExpLo) < c.span;
}

__device__ __forceinline__ double div_const(double n, const ConstDiv& c)
{
    const double q = n * c.r;
    const double rem = __builtin_fma(-c.d, q, n);
    double res = __builtin_fma(rem, c.r, q);
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

__device__ __forceinline__ bool is_finite(double x)
{
    return __builtin_isfinite(x);
}

}  // namespace rscm
