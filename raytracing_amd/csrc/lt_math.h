// Exact integer arithmetic of the polygon path as device functions: wrapping 32-bit
// operations and the divisions, which go through the floating-point units (the integer
// divide sequence is several times slower).  Each function states the operand range it is
// exact for; tests/csrc/lt_kat.hip evaluates them on arrays for the known-answer tests.
#pragma once
#include <hip/hip_runtime.h>

namespace lt {

__device__ __forceinline__ int wadd(int a, int b) { return (int)((unsigned)a + (unsigned)b); }
__device__ __forceinline__ int wsub(int a, int b) { return (int)((unsigned)a - (unsigned)b); }
__device__ __forceinline__ int wmul(int a, int b) { return (int)((unsigned)a * (unsigned)b); }
__device__ __forceinline__ int wshl(int a, int s) { return (int)((unsigned)a << s); }
// truncating signed division; b != 0.  INT_MIN / -1 wraps to INT_MIN.
__device__ __forceinline__ int tdiv(int a, int b)
{
    if (b == -1) return (int)(0u - (unsigned)a);
    return a / b;
}

// Exact truncating division through the double-precision unit, several times cheaper
// than the 32-bit integer divide sequence.  For integers n and d >= 1,
// trunc(n/d) = sign(n) * floor((|n| + 1/2) / d): the half never reaches the next
// multiple of d, and it keeps the product at least 1/(2d) away from every integer, far
// more than the rounding error of the multiply (|n| * 2^-51 / d for |n| < 2^50).
// drcp(d) is 1/d to double precision (v_rcp_f64 + two Newton steps), shared by all
// numerators that have the same divisor.
__device__ __forceinline__ double drcp_d(double x)
{
    double r = __builtin_amdgcn_rcp(x);
    r = fma(fma(-x, r, 1.0), r, r);
    r = fma(fma(-x, r, 1.0), r, r);
    return r;
}
__device__ __forceinline__ double drcp(int d) { return drcp_d((double)d); }
__device__ __forceinline__ int tdivr(int n, double r)
{
    const double h = n < 0 ? -0.5 * r : 0.5 * r;
    return (int)fma((double)n, r, h);
}

// floor(n / d) for 0 <= n < 2^30 and 1 <= d <= 2^15 with a quotient below 2^15 (the
// point-index divisions of the edge walk), given rd ~ 1/d in single precision: the
// estimate is within one of the quotient, and one exact correction step from the
// remainder settles it.  The products fit 24-bit multiplies (full rate on the VALU).
__device__ __forceinline__ int udiv_r(int n, int d, float rd)
{
    int q = (int)((float)n * rd);
    int r = n - __mul24(q, d);
    q = r < 0 ? q - 1 : q;
    r = r < 0 ? r + d : r;
    return r >= d ? q + 1 : q;
}

// TI_sqrt, TransLn.cpp:38-51: sixteen result bits, 64-bit accumulators (long is 64 bits in
// the pinned build).  The bit-by-bit loop yields floor(sqrt(arg)) while that fits sixteen
// bits and 0xffff beyond; here through the f64 square root with an exact integer fix-up.
__device__ __forceinline__ unsigned long long ti_sqrt(unsigned long long arg)
{
    if (arg >= 0xffffull * 0xffffull) return 0xffffull;
    unsigned r = (unsigned)__builtin_sqrt((double)arg);          // arg < 2^32: exact in double
    r = (unsigned long long)r * r > arg ? r - 1u : r;
    r = (unsigned long long)(r + 1u) * (r + 1u) <= arg ? r + 1u : r;
    return r;
}

// (v << 7) / lng of T_unit_vector (TransLn.cpp:186-188), truncating.  |v << 7| < 2^39 and
// 0 < lng <= 2^20, both exact in double; a quotient that is not an integer is at least
// 1/lng away from one, far more than the rounding of the division, so the truncated f64
// quotient is the integer quotient.
__device__ __forceinline__ int unit_component(int v, long long lng)
{
    return (int)(long long)(((double)v * 128.0) / (double)lng);
}

// T_perspective (TransBs.cpp:425-429): (int)(((long)x << log_focus) / z).  The 64-bit
// quotient is only needed when x << log_focus leaves 32 bits.
__device__ __forceinline__ int persp_div(int x, int lf, int z, double rz)
{
    int xs = wshl(x, lf);
    if ((xs >> lf) == x) return tdivr(xs, rz);
    return (int)(((long long)x * ((long long)1 << lf)) / (long long)z);
}

// truncating division for any non-zero divisor through the f64 unit
__device__ __forceinline__ int tdiv_any(int n, int d)
{
    const bool neg = d < 0;
    const int q = tdivr(n, drcp_d(neg ? -(double)d : (double)d));        // |d| <= 2^31 is exact in double (not in int)
    return neg ? (int)(0u - (unsigned)q) : q;
}

}  // namespace lt
