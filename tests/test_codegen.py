"""Code-generation guard (no GPU): the uniform loads of render_kernel -- quads, spheres, materials, the root of the tree -- must go
through the scalar cache.  The compiler gives a load to the scalar unit only while it can prove that nothing in the kernel writes
before it; one side-effecting intrinsic in front of the frame loop (a time stamp, an `asm volatile`, an LDS atomic) turns every one of
them into a vector load -- bit-exact, 11 ... 19 % slower, and invisible to every parity test ("A time stamp
de-scalarises the kernel").  This compiles the headline material set to a listing and counts."""
import os
import re
import shutil
import subprocess

import pytest

from conftest import ROOT

HIP = os.path.join(ROOT, "photorealistic-rendering-using-opencl_amd", "csrc", "hip")
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


def _load_counts(tmp_path, name, flags=()):
    """per render_kernel instance of the set: [dwords fetched by scalar loads, vector (global) load instructions]"""
    out = tmp_path / (name + "".join(flags) + ".s")
    cmd = [HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-fast-math", "-fno-slp-vectorize", "-S",
           "--cuda-device-only", "-I" + os.path.join(ROOT, "include"), "-I" + HIP] + list(flags) + ["-o", str(out), os.path.join(HIP, "pt_inst_%s.hip" % name)]
    subprocess.run(cmd, check=True, capture_output=True, timeout=1800)
    kernels, cur = {}, None
    for line in out.read_text().split("\n"):
        m = re.match(r"^(_ZN3prt13render_kernel\w+):", line)
        if m:
            cur = m.group(1)
            kernels[cur] = [0, 0]
        elif line.startswith(".Lfunc_end"):
            cur = None
        elif cur:
            m = re.match(r"\s+s_load_dword(x(\d+))?\s", line)
            if m:
                kernels[cur][0] += int(m.group(2) or 1)
            elif re.match(r"\s+global_load", line):
                kernels[cur][1] += 1
    return kernels


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="no hipcc")
def test_uniform_loads_of_the_render_kernel_stay_scalar(tmp_path):
    """every material set's listing: the scalar unit fetches 240 ... 620 dwords per kernel (the quads, spheres, materials and the root
    pair as dwordx4 / x8 loads through the constant address space, pt_device.h PT_CONST) against 36 ... 70 vector load instructions;
    the round-3 kernel de-scalarised (measured on the headline set): 140 dwords against 76 vector loads"""
    from concurrent.futures import ThreadPoolExecutor
    names = sorted(f[len("pt_inst_"):-len(".hip")] for f in os.listdir(HIP) if f.startswith("pt_inst_") and f.endswith(".hip"))
    assert "light_diff" in names and len(names) >= 10, names
    with ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 1)) as pool:
        results = dict(zip(names, pool.map(lambda n: _load_counts(tmp_path, n), names)))
    for name, kernels in results.items():
        assert len(kernels) >= 3, (name, kernels)               # the wave-count builds (x medium off / on where the set has both)
        for k, (scalar_dw, vector) in kernels.items():
            assert scalar_dw >= 200 and vector <= 75, (name, k, scalar_dw, vector)
    for k, (scalar_dw, vector) in results["light_diff"].items():
        assert scalar_dw >= 240 and vector <= 45, (k, scalar_dw, vector)


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="no hipcc")
def test_uniform_loads_stay_scalar_behind_a_write(tmp_path):
    """Scalar BY CONSTRUCTION (round 4): -DPT_TEST_CLOBBER puts an `asm volatile("" ::: "memory")` and an LDS atomic in front of the frame
    loop of render_kernel -- what a time stamp or a ray pool's queue is to the compiler.  With the round-3 sources that turned the
    headline set's 254 scalar dwords into 138 and its 38 vector loads into 76; through the constant address space nothing moves."""
    clean = _load_counts(tmp_path, "light_diff")
    dirty = _load_counts(tmp_path, "light_diff", ("-DPT_TEST_CLOBBER",))
    assert set(clean) == set(dirty) and len(clean) >= 3
    for k in clean:
        assert dirty[k][0] >= 240 and dirty[k][0] >= clean[k][0], (k, clean[k], dirty[k])     # no scalar load lost ...
        assert dirty[k][1] <= clean[k][1], (k, clean[k], dirty[k])                              # ... and no vector load gained
