"""The device math header compiles for the host too: check og_sinf / og_tanf_q1 against glibc
(the libm the reference's f32::sin / f32::tan bind to on Linux).  CPU only."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = r'''
#include "og_math.h"
extern "C" void og_sin_array(const float* x, float* y, long n) { for (long i = 0; i < n; ++i) y[i] = og_sinf(x[i]); }
extern "C" void og_sin_turns_array(const float* x, float* y, long n) { for (long i = 0; i < n; ++i) y[i] = og_sin_turns_poly(x[i]); }
extern "C" void og_sin_turns_dflt(const float* x, float* y, long n) { for (long i = 0; i < n; ++i) y[i] = og_sin_turns(x[i]); }
extern "C" void og_tan_array(const float* x, float* y, long n) { for (long i = 0; i < n; ++i) y[i] = og_tanf_q1(x[i]); }
extern "C" void ex_sin_array(const float* x, float* y, long n) { for (long i = 0; i < n; ++i) y[i] = og_sinf_exact(x[i]); }
extern "C" void ex_cos_array(const float* x, float* y, long n) { for (long i = 0; i < n; ++i) y[i] = og_cosf_exact(x[i]); }
extern "C" void ref_cos_array(const float* x, float* y, long n) { for (long i = 0; i < n; ++i) y[i] = cosf(x[i]); }
extern "C" void ref_sin_array(const float* x, float* y, long n) { for (long i = 0; i < n; ++i) y[i] = sinf(x[i]); }
extern "C" void ref_tan_array(const float* x, float* y, long n) { for (long i = 0; i < n; ++i) y[i] = tanf(x[i]); }
'''


@pytest.fixture(scope="module", params=["tolerance", "strict"])
def mlib(tmp_path_factory, request):
    # "tolerance" = the shipped mode (4-coefficient sine polynomial), "strict" = -DOG_STRICT (rounds 1-3)
    d = tmp_path_factory.mktemp("ogmath_" + request.param)
    src = d / "m.cpp"
    src.write_text(SRC)
    so = d / "libm_test.so"
    flags = ["-O2", "-ffp-contract=off", "-shared", "-fPIC", "-I" + os.path.join(ROOT, "oscen_amd", "csrc")]
    if request.param == "strict":
        flags.append("-DOG_STRICT")
    if "fma" in open("/proc/cpuinfo").read():
        flags.append("-mfma")
    subprocess.run(["g++"] + flags + [str(src), "-o", str(so), "-lm"], check=True)
    return C.CDLL(str(so))


def _apply(lib, name, x):
    y = np.empty_like(x)
    fp = C.POINTER(C.c_float)
    getattr(lib, name)(x.ctypes.data_as(fp), y.ctypes.data_as(fp), C.c_long(len(x)))
    return y


def test_sin_matches_glibc_over_fm_range(mlib):
    rng = np.random.default_rng(0)
    x = np.concatenate([np.linspace(-64, 64, 2_000_001), rng.uniform(-8192, 8192, 500_000),
                        rng.uniform(-1e5, 1e5, 200_000), np.array([0.0, -0.0, 1e-30, np.pi, -np.pi])]).astype(np.float32)
    got, ref = _apply(mlib, "og_sin_array", x), _apply(mlib, "ref_sin_array", x)
    assert np.max(np.abs(got.astype(np.float64) - ref)) <= 2.4e-7
    small = np.abs(x) <= 64
    assert np.max(np.abs(got[small].astype(np.float64) - ref[small])) <= 1.2e-7
    assert np.mean(got[small] == ref[small]) > 0.70  # the rest differ by 1 ulp
    true = np.sin(x[small].astype(np.float64))
    # (tolerance mode drops the degree-11 term: approximation error 4.7e-9 on top of the final rounding)
    assert np.max(np.abs(got[small] - true)) <= 1.45e-7


def test_sin_in_turns(mlib):
    """og_sin_turns_poly(t) = sin(2 pi t): the half-turn reduction is exact, so the error against the TRUE sine is the
    polynomial's (4.9e-9) plus the final roundings at ANY argument size; against the reference's f32
    `((t) * TAU).sin()` the two differ by the rounding of the reference's own product (|t| * 2 pi * 6e-8).  On the GPU the
    shipped operator takes v_sin_f32 (OG_SIN_TURNS = 2; measured on MI355X: 4.6e-7 / 7.6e-7 / 1.5e-6 against the
    reference's form at modulation depths 0 / 1 / 4 turns, profiles/r05a_session1.log); host builds of og_sin_turns
    take this polynomial (strict builds: og_sinf(t * TAU))."""
    rng = np.random.default_rng(2)
    for depth, bound_ref in ((0.0, 4.5e-7), (1.0, 8e-7), (4.0, 1.6e-6), (16.0, 3.7e-6)):
        t = (rng.random(1_000_000) + (rng.random(1_000_000) - 0.5) * depth).astype(np.float32)
        got = _apply(mlib, "og_sin_turns_array", t).astype(np.float64)
        true = np.sin(2.0 * np.pi * t.astype(np.float64))
        assert np.max(np.abs(got - true)) <= 1.6e-7
        x = (t * np.float32(6.28318548202514648)).astype(np.float32)
        assert np.max(np.abs(got - _apply(mlib, "ref_sin_array", x))) <= bound_ref
    big = rng.uniform(-2.0e6, 2.0e6, 200_000).astype(np.float32)  # |2t| < 2^22: the magic-number rounding still holds
    assert np.max(np.abs(_apply(mlib, "og_sin_turns_array", big) - np.sin(2.0 * np.pi * big.astype(np.float64)))) <= 1.6e-7
    exact = np.array([0.0, 0.5, 1.0, -0.5, 2.5, 0.25, -0.25, 0.75], dtype=np.float32)
    want = np.array([0.0, 0.0, 0.0, 0.0, 0.0, 1.0, -1.0, -1.0])
    assert np.max(np.abs(_apply(mlib, "og_sin_turns_array", exact) - want)) <= 1.2e-7
    # the dispatching form: strict builds keep the radian polynomial on the reference's product
    t = rng.random(100_000).astype(np.float32)
    d = _apply(mlib, "og_sin_turns_dflt", t)
    assert np.max(np.abs(d - np.sin(2.0 * np.pi * t.astype(np.float64)))) <= 5e-7


def test_tan_matches_glibc_on_first_quadrant(mlib):
    x = np.linspace(1e-4, 1.5707, 2_000_001).astype(np.float32)
    got, ref = _apply(mlib, "og_tan_array", x), _apply(mlib, "ref_tan_array", x)
    rel = np.abs(got.astype(np.float64) - ref) / np.abs(ref)
    assert rel.max() <= 2.5e-7


def test_exact_sincos_is_bit_identical_to_glibc(mlib):
    # the restated glibc/ARM sincosf algorithm must reproduce the host libm bit for bit
    # (glibc picks its FMA build on AVX2+FMA CPUs; the restatement follows that variant)
    if "fma" not in open("/proc/cpuinfo").read():
        pytest.skip("host CPU without FMA: glibc uses its non-FMA sincosf build")
    rng = np.random.default_rng(1)
    x = np.concatenate([np.linspace(0.0, np.pi, 3_000_001), rng.uniform(-100.0, 100.0, 20_000_000),
                        np.geomspace(1e-8, 119.0, 500_000), np.array([0.0, -0.0, np.pi / 4, -np.pi / 4])]).astype(np.float32)
    assert np.array_equal(_apply(mlib, "ex_sin_array", x), _apply(mlib, "ref_sin_array", x))
    assert np.array_equal(_apply(mlib, "ex_cos_array", x), _apply(mlib, "ref_cos_array", x))
