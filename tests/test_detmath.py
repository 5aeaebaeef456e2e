"""include/prt_detmath.h: accuracy of the host evaluation against float64 libm (so it is a
legitimate OpenCL built-in library) and, on the GPU, bit-identity of device and host evaluation
(the numerics contract everything else rests on)."""
import ctypes as C
import os

import numpy as np
import pytest

from conftest import ROOT

FN = {"sin": 0, "cos": 1, "tan": 2, "exp": 3, "log": 4, "acos": 5, "atan2": 6, "pow": 7, "sqrt": 8, "div": 9,
      "fma": 10, "fmin": 11, "fmax": 12, "round": 13, "floor": 14, "recip": 15, "cbrt": 16}


@pytest.fixture(scope="module")
def probe(oracle):
    lib = C.CDLL(os.path.join(ROOT, "oracle", "libdetmath_probe.so"))
    lib.detmath_probe.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]

    def run(fn, a, b=None):
        a = np.ascontiguousarray(a, dtype=np.float32)
        b = np.zeros_like(a) if b is None else np.ascontiguousarray(b, dtype=np.float32)
        out = np.zeros_like(a)
        lib.detmath_probe(FN[fn], a.ctypes.data, b.ctypes.data, out.ctypes.data, a.size)
        return out
    return run


def ulp_err(got, ref64):
    ref32 = ref64.astype(np.float32)
    ulp = np.spacing(np.abs(ref32)).astype(np.float64)
    return np.abs(got.astype(np.float64) - ref64) / ulp


def test_accuracy_against_libm(probe):
    rng = np.random.default_rng(7)
    x = rng.uniform(-7, 7, 400000).astype(np.float32)
    assert ulp_err(probe("sin", x), np.sin(x.astype(np.float64))).max() < 2.0
    assert ulp_err(probe("cos", x), np.cos(x.astype(np.float64))).max() < 2.0
    big = rng.uniform(-1e4, 1e4, 400000).astype(np.float32)
    assert ulp_err(probe("sin", big), np.sin(big.astype(np.float64))).max() < 2.0
    t = rng.uniform(-1.5, 1.5, 400000).astype(np.float32)
    assert ulp_err(probe("tan", t), np.tan(t.astype(np.float64))).max() < 4.0
    e = rng.uniform(-87, 88, 400000).astype(np.float32)
    assert ulp_err(probe("exp", e), np.exp(e.astype(np.float64))).max() < 2.0
    l = np.exp(rng.uniform(-80, 80, 400000)).astype(np.float32)
    assert ulp_err(probe("log", l), np.log(l.astype(np.float64))).max() < 2.0
    a = rng.uniform(-1, 1, 400000).astype(np.float32)
    assert ulp_err(probe("acos", a), np.arccos(a.astype(np.float64))).max() < 3.0
    y, xx = rng.uniform(-10, 10, 400000).astype(np.float32), rng.uniform(-10, 10, 400000).astype(np.float32)
    assert ulp_err(probe("atan2", y, xx), np.arctan2(y.astype(np.float64), xx.astype(np.float64))).max() < 3.0
    cb = np.concatenate([rng.uniform(-30, 30, 200000), np.exp(rng.uniform(-100, 85, 200000))]).astype(np.float32)
    assert ulp_err(probe("cbrt", cb), np.cbrt(cb.astype(np.float64))).max() < 2.0
    p = rng.uniform(0.01, 3, 100000).astype(np.float32)
    assert np.array_equal(probe("pow", p, np.full_like(p, 2.0)), p * p)                # exact
    q = rng.uniform(-4, 4, 100000).astype(np.float32)
    # general pow is computed in binary64 and rounded once (OpenCL allows 16 ulp; round 1-2's exp(y log x) in binary32 reached tens)
    assert ulp_err(probe("pow", p, q), np.power(p.astype(np.float64), q.astype(np.float64))).max() < 0.51
    # the arguments of the Phong lobe (kernels/bxdf/microfacet.cl:31-33,95-97): exponents 2 / r^2 - 2 up to 2e6 and their reciprocals
    c = rng.uniform(0, 1, 200000).astype(np.float32)
    rr = rng.uniform(1e-3, 1, 200000).astype(np.float32)
    alpha = (np.float32(2.0) / (rr * rr) - np.float32(2.0)).astype(np.float32)
    want = np.power(c.astype(np.float64), alpha.astype(np.float64))
    ok = want > 1.2e-38                                                            # normal results
    assert ulp_err(probe("pow", c, alpha)[ok], want[ok]).max() < 0.51
    inv = (np.float32(1.0) / (alpha + np.float32(2.0))).astype(np.float32)
    assert ulp_err(probe("pow", c, inv), np.power(c.astype(np.float64), inv.astype(np.float64))).max() < 0.51


def test_special_values(probe):
    one = np.array([0.0, -0.0], np.float32)
    assert np.array_equal(probe("exp", one), [1.0, 1.0])
    assert probe("log", np.array([1.0], np.float32))[0] == 0.0
    assert np.isneginf(probe("log", np.array([0.0], np.float32))[0]) and np.isnan(probe("log", np.array([-1.0], np.float32))[0])
    assert probe("acos", np.array([1.0], np.float32))[0] == 0.0
    assert np.array_equal(probe("round", np.array([2.5, -0.5, 0.49999997, -2.5, 1e10], np.float32)), np.array([3, -1, 0, -3, 1e10], np.float32))
    assert np.array_equal(probe("floor", np.array([-1.5, 1.5, -0.0, 8388609.0], np.float32)), np.array([-2, 1, -0.0, 8388609.0], np.float32))
    nan = np.float32(np.nan)
    assert probe("fmin", np.array([nan, 1.0], np.float32), np.array([2.0, nan], np.float32)).tolist() == [2.0, 1.0]
    assert probe("fmax", np.array([nan, 1.0], np.float32), np.array([2.0, nan], np.float32)).tolist() == [2.0, 1.0]
    z = probe("fmax", np.array([0.0, -0.0], np.float32), np.array([-0.0, 0.0], np.float32))
    assert np.signbit(z).tolist() == [False, True]                                      # ties return the first operand
    assert np.isnan(probe("sin", np.array([np.inf], np.float32))[0])


def edge_inputs():
    rng = np.random.default_rng(11)
    specials = np.array([0.0, -0.0, 1.0, -1.0, 0.5, -0.5, 2.0, np.inf, -np.inf, np.nan, 1e-45, -1e-45, 1.1754944e-38,
                         3.4028235e38, -3.4028235e38, 1e-5, 20.0, 88.7, -103.9, -104.0, 8388608.0, 0.49999997, 2.5, -2.5,
                         3.14159274, 6.28318548, 1e4, 1e9, 2e9], np.float32)
    a = np.concatenate([specials, rng.uniform(-10, 10, 200000).astype(np.float32),
                        np.exp(rng.uniform(-90, 90, 50000)).astype(np.float32), rng.uniform(0, 1, 100000).astype(np.float32)])
    b = np.concatenate([specials[::-1], rng.uniform(-10, 10, 200000).astype(np.float32),
                        rng.uniform(-3, 3, 50000).astype(np.float32), rng.uniform(0, 1, 100000).astype(np.float32)])
    return a, b


@pytest.mark.gpu
def test_device_equals_host_bit_for_bit(prt, oracle, probe):
    scene = prt.HostScene("cornell_diffuse.json")
    r = prt.Renderer(scene.config(), device=0)
    a, b = edge_inputs()
    for name, fn in FN.items():
        dev = r.selftest_math(fn, a, b)
        host = probe(name, a, b)
        same = oracle.float_bits(dev) == oracle.float_bits(host)
        bad = np.flatnonzero(~same)
        assert bad.size == 0, "%s differs at %d inputs, e.g. a=%r b=%r dev=%r host=%r" % (
            name, bad.size, a[bad[0]], b[bad[0]], dev[bad[0]], host[bad[0]])
    r.close()
