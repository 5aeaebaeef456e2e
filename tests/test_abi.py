"""C-ABI surface: libprt.so loads, exports every symbol include/prt.h declares, the struct layouts
are the reference's, and the product fails loudly (never falls back) without a HIP device."""
import ctypes as C
import os
import re

import pytest

from conftest import ROOT


def test_every_declared_symbol_is_exported(prt):
    header = open(os.path.join(ROOT, "include", "prt.h")).read()
    declared = set(re.findall(r"\b(prt_[a-z_0-9]+)\s*\(", header))
    declared -= {"prt_ctx"}
    assert len(declared) >= 20
    lib = C.CDLL(os.path.join(ROOT, prt.__name__, "libprt.so"))
    for name in sorted(declared):
        assert hasattr(lib, name), "libprt.so does not export %s" % name
    bound = {n for n, _, _ in prt._capi.PRT_API}
    assert declared == bound, "python prototypes out of sync with include/prt.h: %s" % (declared ^ bound)


def test_struct_layouts_match_the_reference_abi(prt):
    c = prt._capi
    assert C.sizeof(c.Material) == 64 and C.sizeof(c.Mesh) == 256 and C.sizeof(c.BvhNode) == 36 and C.sizeof(c.Camera) == 80
    assert c.Mesh.joker.offset == 128 and c.Mesh.t.offset == 192 and c.Mesh.pos.offset == 64
    assert c.Material.roughness.offset == 48 and c.Material.t.offset == 52 and c.Material.lobes.offset == 54 and c.Material.dist.offset == 55
    assert c.Camera.resolution.offset == 48 and c.Camera.fov.offset == 56 and c.Camera.apertureRadius.offset == 64
    import numpy as np
    dt = np.dtype(c.PATH_STATE_DTYPE)
    assert dt.itemsize == 112
    assert dt.fields["mask"][1] == 48 and dt.fields["acc"][1] == 64 and dt.fields["total"][1] == 80
    assert dt.fields["was_specular"][1] == 92 and dt.fields["reset"][1] == 96 and dt.fields["samples"][1] == 100


def test_no_silent_fallback_without_a_device(prt):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    scene = prt.HostScene("cornell_diffuse.json")
    with pytest.raises(prt.PrtError) as ei:
        prt.Renderer(scene.config(), device=0)
    assert "no HIP device" in str(ei.value) or "no CPU fallback" in str(ei.value)


def test_product_does_not_reference_the_oracle(prt):
    """the product path must not import, link or call anything under oracle/"""
    pkg = os.path.join(ROOT, prt.__name__)
    for d, _, files in os.walk(pkg):
        if os.path.basename(d) in ("build", "variants", "__pycache__"):
            continue
        for f in files:
            if f.endswith((".py", ".cpp", ".h", ".hip")):
                text = open(os.path.join(d, f)).read()
                for needle in ("oracle_api", "pt_oracle", "liboracle", "pto_render"):
                    if needle in text and not (f.endswith((".h", ".hip", ".cpp")) and "oracle/" in text and needle == "pt_oracle"):
                        assert False, "%s mentions %s" % (os.path.join(d, f), needle)
    import subprocess
    out = subprocess.run(["ldd", os.path.join(pkg, "libprt.so")], stdout=subprocess.PIPE, text=True).stdout
    assert "oracle" not in out


def test_the_library_says_what_it_was_built_from_and_the_bench_refuses_a_stale_one(prt):
    """prt_build_id(): build.py hashes the content of every source, header and flag into the library.  Touching a header WITHOUT
    rebuilding must be seen: the package's check raises, and bench.py leaves with a non-zero status and one line on stderr before it
    looks for a GPU (round 3 timed an experiment's left-over library for a day, twice)."""
    import subprocess
    import sys
    assert prt.build_id() == prt.source_build_id() and len(prt.build_id()) == 16
    prt.check_build_id()
    header = os.path.join(ROOT, "include", "prt_types.h")
    st = os.stat(header)
    original = open(header, "rb").read()
    env = {k: v for k, v in os.environ.items() if k != "PRT_LIB"}
    try:
        with open(header, "ab") as fh:
            fh.write(b"\n/* touched by tests/test_abi.py */\n")
        os.utime(header, ns=(st.st_atime_ns, st.st_mtime_ns))              # (no rebuild may be triggered by this test: same mtime)
        assert prt.source_build_id() != prt.build_id()
        with pytest.raises(prt.StaleLibrary):
            prt.check_build_id()
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "0"], capture_output=True, text=True, env=env, timeout=600)
        assert r.returncode == 3, (r.returncode, r.stderr[-500:])
        assert "refusing to time a stale library" in r.stderr and not r.stdout.strip(), (r.stdout, r.stderr[-500:])
    finally:
        with open(header, "wb") as fh:
            fh.write(original)
        os.utime(header, ns=(st.st_atime_ns, st.st_mtime_ns))
    prt.check_build_id()
