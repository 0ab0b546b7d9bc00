// og_math.h -- accurate, branch-light f32 transcendental functions for the
// voice kernels (gfx950).  Plain C++ with explicit fmaf so the very same code
// compiles for the host, where tests/test_og_math.py checks it against glibc
// (the libm the reference's Rust `f32::sin/tan` bind to on Linux).
//
// The reference computes e.g. `((phase + mod) * TAU).sin()`
// (examples/fm-synth/src/nodes/fm_operator.rs:63-64).  The argument is formed
// with the reference's exact f32 ops by the caller; these functions only
// replace the libm call, with abs error <= ~1.5e-7 (sin) so that a 3-operator
// FM chain (phase sensitivity 2*pi per operator) stays inside the 1e-5 parity
// budget.  The hardware v_sin_f32 is deliberately not used: its absolute error
// is too large once amplified through two modulation stages.
#pragma once
#ifndef __HIPCC_RTC__
#include <math.h>
#include <stdint.h>
#else
using __hip_internal::int32_t;
using __hip_internal::uint32_t;
#endif

#if defined(__HIPCC__)
#define OG_HD __host__ __device__ __forceinline__
#else
#define OG_HD static inline
#endif

// sin(x): Cody-Waite reduction modulo pi with the 3-term split of pi carried
// by fma (k*PI_A is exact inside the fma, so the reduction stays accurate for
// |x| up to ~1e6 -- the graphs in scope keep |phase + mod| below ~8 turns),
// then an odd minimax polynomial on [-pi/2, pi/2] (Remez fit of
// (sin r - r)/r^3 in r^2, 5 coefficients, max error 3.4e-11 * r^3).
// No slow path: beyond ~1e6 the result degrades with ulp(x) like any f32 sin.

OG_HD float og_sin_reduced(float r)
{
    const float S1 = -0x1.555556p-3f;   // -0.16666667
    const float S2 = 0x1.111110p-7f;    //  0.0083333328
    const float S3 = -0x1.a018e8p-13f;  // -0.00019841064
    const float S4 = 0x1.7190e0p-19f;   //  2.7534806e-06
    const float S5 = -0x1.9d0bc6p-26f;  // -2.4042441e-08
    float u = r * r;
    float q = fmaf(u, S5, S4);
    q = fmaf(u, q, S3);
    q = fmaf(u, q, S2);
    q = fmaf(u, q, S1);
    float r3 = u * r;
    return fmaf(r3, q, r);
}

OG_HD float og_sinf(float x)
{
    const float INV_PI = 0x1.45f306p-2f;   // 1/pi
    const float PI_A = 0x1.921fb6p+1f;     // fl(pi)
    const float PI_B = -0x1.777a5cp-24f;   // fl(pi - PI_A)
    const float PI_C = -0x1.ee59dap-49f;   // fl(pi - PI_A - PI_B)
    float k = rintf(x * INV_PI);
    float r = fmaf(-k, PI_A, x);
    r = fmaf(-k, PI_B, r);
#ifdef OG_SIN_3TERM
    r = fmaf(-k, PI_C, r);
#else
    (void)PI_C; // |k| * 3.4e-15: below half an ulp of r for |x| < 1e5
#endif
    // (-1)^k: flip the sign of r (odd polynomial) when k is odd
    int32_t ki = (int32_t)k;
    union { float f; uint32_t u; } b;
    b.f = r;
    b.u ^= ((uint32_t)ki) << 31;
    return og_sin_reduced(b.f);
}

// tan(x) for x in [0, pi/2): used by the TPT coefficient update
// (oscen-lib/src/filters/tpt/mod.rs:73; argument = pi * fc / sr).
// x <= pi/4: odd minimax polynomial; otherwise 1 / tan(pi/2 - x) with the
// complement formed exactly (Sterbenz) plus the low word of pi/2.
OG_HD float og_tan_poly(float y)
{
    const float T1 = 0x1.555556p-2f;
    const float T2 = 0x1.111060p-3f;
    const float T3 = 0x1.ba5f56p-5f;
    const float T4 = 0x1.61821cp-6f;
    const float T5 = 0x1.4b49a4p-7f;
    const float T6 = 0x1.f7a1aep-11f;
    const float T7 = 0x1.043870p-8f;
    float u = y * y;
    float q = fmaf(u, T7, T6);
    q = fmaf(u, q, T5);
    q = fmaf(u, q, T4);
    q = fmaf(u, q, T3);
    q = fmaf(u, q, T2);
    q = fmaf(u, q, T1);
    return fmaf(u * y, q, y);
}

OG_HD float og_tanf_q1(float x)
{
    const float PIO4 = 0x1.921fb6p-1f;
    const float PIO2_HI = 0x1.921fb6p+0f;
    const float PIO2_LO = -0x1.777a5cp-25f;
    if (x <= PIO4) return og_tan_poly(x);
    float y = (PIO2_HI - x) + PIO2_LO;
    return 1.0f / og_tan_poly(y);
}
