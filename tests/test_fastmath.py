"""CPU validation of csrc/agp_math.hpp (the fp64 exp / sin^2 / pow used by the covariance kernels):
the header is host+device, so it is compiled here with g++ and checked against mpmath."""
import ctypes
import subprocess
from pathlib import Path

import mpmath as mp
import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent
SRC = r'''
#include "agp_math.hpp"
extern "C" {
void v_exp(const double* x, double* y, int n) { for (int i = 0; i < n; ++i) y[i] = agp::fm::exp_f(x[i]); }
void v_exp_t(const double* x, double* y, int n) { for (int i = 0; i < n; ++i) y[i] = agp::fm::exp_t(x[i], agp::fm::EXP_TAB); }
void v_sin2(const double* x, double* y, int n) { for (int i = 0; i < n; ++i) y[i] = agp::fm::sin2_f(x[i]); }
void v_sincos(const double* x, double* y, int n) { for (int i = 0; i < n; ++i) agp::fm::sincos_pi_f(x[i], y + 2 * i, y + 2 * i + 1); }
void v_log(const double* x, double* y, int n) { for (int i = 0; i < n; ++i) y[i] = agp::fm::log_f(x[i]); }
void v_pow(const double* x, const double* g, double* y, int n) { for (int i = 0; i < n; ++i) y[i] = agp::fm::pow_f(x[i], g[i]); }
}
'''


@pytest.fixture(scope="module")
def lib(tmp_path_factory):
    d = tmp_path_factory.mktemp("fm")
    (d / "fm.cpp").write_text(SRC)
    so = d / "libfm.so"
    subprocess.run(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared", "-I", str(ROOT / "autogp.jl_amd" / "csrc"),
                    "-o", str(so), str(d / "fm.cpp")], check=True)
    return ctypes.CDLL(str(so))


def call1(fn, x):
    x = np.ascontiguousarray(x, dtype=np.float64); y = np.empty_like(x)
    fn(x.ctypes.data_as(ctypes.c_void_p), y.ctypes.data_as(ctypes.c_void_p), ctypes.c_int(x.size))
    return y


def ulps(got, ref):
    """error in units of the last place of the reference"""
    ref_f = np.array([float(r) for r in ref])
    err = np.array([abs(mp.mpf(float(g)) - r) for g, r in zip(got, ref)], dtype=object)
    return np.array([float(e / mp.mpf(np.spacing(abs(rf)) if rf != 0 else 5e-324)) for e, rf in zip(err, ref_f)])


def test_exp(lib):
    mp.mp.dps = 40
    rng = np.random.default_rng(0)
    x = np.concatenate([-rng.random(1500) * 50, -np.exp(rng.uniform(-40, 6.5, 1500)), rng.random(500) * 12,
                        [0.0, -1e-300, -700.0, -1e-17, 11.5]])
    u = ulps(call1(lib.v_exp, x), [mp.exp(mp.mpf(float(v))) for v in x])
    assert u.max() < 1.6, u.max()
    assert call1(lib.v_exp, np.array([-746.0, -1000.0, -1e9]))[0:3].tolist() == [0.0, 0.0, 0.0]
    sub = call1(lib.v_exp, np.array([-720.0]))[0]          # subnormal result
    assert abs(sub - float(mp.exp(-720))) <= 5e-324 * 2


def test_exp_table(lib):
    """exp_t (128-entry table of 2^(j/128) + degree-5 polynomial: the version the covariance evaluation uses on the device):
    same accuracy class as exp_f on the same domain, same underflow behaviour; the table itself is correctly rounded."""
    mp.mp.dps = 40
    rng = np.random.default_rng(1)
    x = np.concatenate([-rng.random(3000) * 50, -np.exp(rng.uniform(-40, 6.5, 3000)), rng.random(800) * 12,
                        np.arange(-1280, 1281) * (np.log(2) / 256),          # ties of the table index
                        [0.0, -1e-300, -700.0, -1e-17, 11.5, 700.0]])
    u = ulps(call1(lib.v_exp_t, x), [mp.exp(mp.mpf(float(v))) for v in x])
    assert u.max() < 1.6, u.max()
    assert call1(lib.v_exp_t, np.array([-746.0, -1000.0, -1e9]))[0:3].tolist() == [0.0, 0.0, 0.0]
    sub = call1(lib.v_exp_t, np.array([-720.0]))[0]
    assert abs(sub - float(mp.exp(-720))) <= 5e-324 * 2
    assert call1(lib.v_exp_t, np.array([0.0]))[0] == 1.0
    # agreement of the two implementations where the kernels use them (arg <= 0)
    a = call1(lib.v_exp, x[x <= 0]); b = call1(lib.v_exp_t, x[x <= 0])
    assert np.max(np.abs(a - b) / np.maximum(a, 1e-300)) < 5e-16


def test_sin2(lib):
    mp.mp.dps = 60
    rng = np.random.default_rng(1)
    x = np.concatenate([rng.random(2000) * 4, rng.random(2000) * 700, rng.random(500) * 1e5, [0.0, 1e-9, np.pi, np.pi / 2]])
    got = call1(lib.v_sin2, x)
    ref = [mp.sin(mp.mpf(float(v))) ** 2 for v in x]
    # near multiples of pi the value is ~0 and only absolute accuracy (relative to 1) is meaningful
    abs_err = np.array([float(abs(mp.mpf(float(g)) - r)) for g, r in zip(got, ref)])
    rel = abs_err / np.maximum(np.array([float(r) for r in ref]), 1e-300)
    good = np.array([float(r) for r in ref]) > 1e-6
    assert rel[good].max() < 1e-15 * 3 and abs_err.max() < 1e-15, (rel[good].max(), abs_err.max())


def test_sincos_pi(lib):
    """sin(x)^2 = s^2 and sin(x) cos(x) = s c after the reduction by pi (Periodic-kernel derivatives)."""
    mp.mp.dps = 60
    rng = np.random.default_rng(5)
    x = np.concatenate([rng.random(2000) * 4, rng.random(2000) * 700, rng.random(500) * 1e5, [0.0, 1e-9, np.pi, np.pi / 2]])
    y = np.empty(2 * x.size)
    lib.v_sincos(x.ctypes.data_as(ctypes.c_void_p), y.ctypes.data_as(ctypes.c_void_p), ctypes.c_int(x.size))
    s, c = y[0::2], y[1::2]
    e2 = max(float(abs(mp.mpf(float(a)) ** 2 - mp.sin(mp.mpf(float(v))) ** 2)) for a, v in zip(s, x))
    esc = max(float(abs(mp.mpf(float(a)) * mp.mpf(float(b)) - mp.sin(mp.mpf(float(v))) * mp.cos(mp.mpf(float(v)))))
              for a, b, v in zip(s, c, x))
    assert e2 < 1e-15 and esc < 1e-15, (e2, esc)


def test_log_pow(lib):
    mp.mp.dps = 40
    rng = np.random.default_rng(2)
    x = np.concatenate([np.exp(rng.uniform(-30, 8, 3000)), [1.0, 0.5, 2.0, 1e-310, 5e-324, 1.4142135623730951, 0.7071067811865476]])
    u = ulps(call1(lib.v_log, x), [mp.log(mp.mpf(float(v))) for v in x])
    assert u[np.abs(x - 1.0) > 1e-3].max() < 1.1
    # pow on the kernels' domain: u = |dx|/l in (0, ~200], gamma in (0, 2]
    uu = np.exp(rng.uniform(-25, 5.3, 4000)); gg = 2.0 / (1.0 + np.exp(-rng.standard_normal(4000)))
    got = np.empty_like(uu)
    lib.v_pow(uu.ctypes.data_as(ctypes.c_void_p), gg.ctypes.data_as(ctypes.c_void_p), got.ctypes.data_as(ctypes.c_void_p),
              ctypes.c_int(uu.size))
    ref = [mp.mpf(float(a)) ** mp.mpf(float(b)) for a, b in zip(uu, gg)]
    rel = np.array([float(abs(mp.mpf(float(g)) - r) / r) for g, r in zip(got, ref)])
    y = np.abs(gg * np.log(uu))
    assert (rel <= (3 + 1.2 * y) * 1.12e-16).all(), (rel / ((3 + 1.2 * y) * 1.12e-16)).max()
    z = np.zeros(3); gz = np.array([0.3, 1.0, 2.0]); out = np.empty(3)
    lib.v_pow(z.ctypes.data_as(ctypes.c_void_p), gz.ctypes.data_as(ctypes.c_void_p), out.ctypes.data_as(ctypes.c_void_p), ctypes.c_int(3))
    assert out.tolist() == [0.0, 0.0, 0.0]


def test_covariance_entries_match_oracle_formulas(lib):
    """End-to-end on the three stationary leaves: entry = amp*exp_f(arg) vs the oracle's libm formula."""
    rng = np.random.default_rng(3)
    dx = np.concatenate([rng.random(4000), [0.0, 1e-12, 1.0]])
    for _ in range(20):
        l, p, amp = np.exp(-1.5 + rng.standard_normal(3)); gam = 2 / (1 + np.exp(-rng.standard_normal()))
        se = amp * call1(lib.v_exp, ((-0.5 * dx) * dx) * (1.0 / (l * l)))
        ref = amp * np.exp(-.5 * dx * dx / l ** 2)
        arg = np.abs(.5 * dx * dx / l ** 2)
        # |arg| ulps of slack: the reciprocal-multiply and the reference's own division round the argument differently
        assert (np.abs(se - ref) <= (3 + 2 * arg) * 1.2e-16 * ref + 1e-300).all()
        s2 = call1(lib.v_sin2, (np.pi / p) * dx)
        per = amp * call1(lib.v_exp, (-2 / l ** 2) * s2)
        ref = amp * np.exp((-2 / l ** 2) * np.sin(np.pi / p * dx) ** 2)
        assert np.abs(per - ref).max() <= 1e-13 * amp
        pw = np.empty_like(dx); g = np.full_like(dx, gam); uu = np.ascontiguousarray(dx * (1.0 / l))
        lib.v_pow(uu.ctypes.data_as(ctypes.c_void_p), g.ctypes.data_as(ctypes.c_void_p), pw.ctypes.data_as(ctypes.c_void_p),
                  ctypes.c_int(dx.size))
        ge = amp * call1(lib.v_exp, -pw)
        assert np.abs(ge - amp * np.exp(-(dx / l) ** gam)).max() <= 1e-13 * amp
