"""Per-function known-answer tests: the REFERENCE's own functions (BSDF2 / BSDF_eval2 / BSDF_pdf for every material and
microfacet distribution, Microfacet_D / G1 / pdf / sample, conductor / dielectric reflectance, sphere and quad light
sampling, HomogeneousMedium_sampleDistance, the Henyey-Greenstein phase function, createCamRay, the primitive tests)
were run on seeded inputs by oracle/ref/kat_harness.cl inside the reference build (tests/golden/make_kat.py); the
fixture tests/golden/kat_functions.npz holds inputs and answers.  Here the product's device functions
(csrc/hip/pt_selftest.h) must give the same bits -- compiled for the host (this file, no GPU) and on the GPU
(tests/test_gpu_parity.py::test_device_functions_match_reference_kat).  End-to-end renders reach rare branches only by
luck; these tables reach them by construction (total internal reflection, the Dirac lobes' eval / pdf, C <= 0 in the
sphere sampler, the quad's back face, grazing microfacet configurations, both camera models)."""
import os
import sys

import numpy as np
import pytest

from conftest import GOLDEN, ROOT

sys.path.insert(0, os.path.join(ROOT, "tests", "emu"))


def kat_tables():
    g = np.load(os.path.join(GOLDEN, "kat_functions.npz"))
    for name in sorted({k.split("/")[0] for k in g.files}):
        yield name, int(g[name + "/fn"]), g[name + "/params"], g[name + "/cases"], g[name + "/expect"], g[name + "/cols"]


def assert_kat_equal(name, got, expect, cols):
    a = np.ascontiguousarray(got[:, cols]).view(np.uint32).copy()
    b = np.ascontiguousarray(expect[:, cols]).view(np.uint32).copy()
    both_nan = np.isnan(got[:, cols]) & np.isnan(expect[:, cols])
    a[both_nan] = 0
    b[both_nan] = 0
    bad = np.argwhere(a != b)
    assert bad.size == 0, "%s: %d of %d cases differ, first: case %d column %d got %r expected %r" % (
        name, len(set(bad[:, 0].tolist())), got.shape[0], bad[0, 0], cols[bad[0, 1]], got[bad[0, 0], cols[bad[0, 1]]], expect[bad[0, 0], cols[bad[0, 1]]])


def test_fixture_covers_the_functions_of_the_path():
    names = [n for n, *_ in kat_tables()]
    assert len(names) >= 40
    for needle in ("bsdf_sample_diff", "bsdf_sample_cond", "bsdf_sample_diel", "bsdf_sample_coat", "bsdf_sample_roughcond_ggx", "bsdf_sample_roughdiel_beckmann",
                   "bsdf_eval_roughdiel_phong", "microfacet_ggx", "fresnel", "light_sphere", "light_quad", "medium", "phase_hg", "camera_pinhole", "hit_quad"):
        assert any(n.startswith(needle) for n in names), needle


def test_device_functions_on_host_match_reference_kat():
    import emu_api
    for name, fn, params, cases, expect, cols in kat_tables():
        assert_kat_equal(name, emu_api.selftest_fn(fn, params, cases), expect, cols)


def _env_cases():
    """environment lookups with answers worked out from the OpenCL 1.2 specification (s8.2, CLK_FILTER_LINEAR with
    normalized coordinates, CLK_ADDRESS_CLAMP = transparent black border, kernels/main.cl:25) in float64: the direction
    maps to (s, t) = (atan2(z, x) / 2pi + 1/2, acos(y) / pi) (kernels/utils.cl:46); u = s w, i0 = floor(u - 1/2),
    a = frac(u - 1/2), likewise v, j0, b; result = (1-a)(1-b) T[i0,j0] + a(1-b) T[i0+1,j0] + (1-a)b T[i0,j0+1] + ab T[i0+1,j0+1]
    with T = 0 outside the image.  Independent of the product's code and of the shim the reference build links."""
    import math
    rng = np.random.default_rng(7)
    tables = []
    for w, h in ((1, 1), (2, 1), (4, 2), (5, 3)):
        tex = rng.uniform(0.1, 4.0, (h, w, 3))
        dirs = []
        for j in range(h):                                     # texel centres: the texel itself, exactly
            for i in range(w):
                s, t = (i + 0.5) / w, (j + 0.5) / h
                phi, theta = (s - 0.5) * 2 * math.pi, t * math.pi
                dirs.append((math.sin(theta) * math.cos(phi), math.cos(theta), math.sin(theta) * math.sin(phi)))
        dirs += [(0, 1, 0), (0, -1, 0), (-1, 0, 0), (-1, 0, 1e-6), (-1, 0, -1e-6), (1, 0, 0)]     # poles, the seam (s = 0 / 1: half border), s = 1/2
        v = rng.normal(size=(40, 3))
        dirs += (v / np.linalg.norm(v, axis=1, keepdims=True)).tolist()
        dirs = np.array(dirs, dtype=np.float32)
        expect = np.zeros((len(dirs), 3))
        for n, d in enumerate(dirs.astype(np.float64)):
            s = math.atan2(d[2], d[0]) / (2 * math.pi) + 0.5
            t = math.acos(max(-1.0, min(1.0, d[1]))) / math.pi
            u, vv = s * w, t * h
            i0, j0 = math.floor(u - 0.5), math.floor(vv - 0.5)
            a, b = (u - 0.5) - i0, (vv - 0.5) - j0
            def T(i, j):
                return tex[j, i] if 0 <= i < w and 0 <= j < h else np.zeros(3)
            expect[n] = (1 - a) * (1 - b) * T(i0, j0) + a * (1 - b) * T(i0 + 1, j0) + (1 - a) * b * T(i0, j0 + 1) + a * b * T(i0 + 1, j0 + 1)
        params = np.zeros(80, dtype=np.float32)
        params[0:2] = np.array([w, h], dtype=np.uint32).view(np.float32)
        params[2:2 + 3 * w * h] = tex.astype(np.float32).reshape(-1)
        cases = np.zeros((len(dirs), 32), dtype=np.float32)
        cases[:, 0:3] = dirs
        tables.append(("%dx%d" % (w, h), params, cases, expect, tex.astype(np.float32)))
    return tables


def check_env_lookup(run):
    for name, params, cases, expect, tex in _env_cases():
        got = run(11, params, cases)[:, 0:3].astype(np.float64)
        # float32 atan2 / acos (<= 3 ulp) move the sample point by ~1e-6 texel: values agree to ~1e-5 of the texel range
        assert np.allclose(got, expect, rtol=0, atol=4e-5 * float(tex.max())), (name, np.abs(got - expect).max())
        n_tex = tex.shape[0] * tex.shape[1]
        centre = got[:n_tex].reshape(tex.shape)
        assert np.allclose(centre, tex, rtol=2e-6, atol=0), name + ": a lookup at a texel centre must return that texel"


def test_env_lookup_follows_the_opencl_sampler_rules_on_host():
    import emu_api
    check_env_lookup(emu_api.selftest_fn)
