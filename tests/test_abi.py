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
