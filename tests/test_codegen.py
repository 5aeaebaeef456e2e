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


def _load_counts(tmp_path, name):
    out = tmp_path / (name + ".s")
    cmd = [HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-fast-math", "-fno-slp-vectorize", "-S",
           "--cuda-device-only", "-I" + os.path.join(ROOT, "include"), "-I" + HIP, "-o", str(out), os.path.join(HIP, "pt_inst_%s.hip" % name)]
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
            if re.match(r"\s+s_load_dword", line):
                kernels[cur][0] += 1
            elif re.match(r"\s+global_load", line):
                kernels[cur][1] += 1
    return kernels


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="no hipcc")
def test_uniform_loads_of_the_render_kernel_stay_scalar(tmp_path):
    """every material set's listing: shipped, 69 ... 143 scalar against 36 ... 64 vector loads per kernel; de-scalarised (measured on the
    headline set): about 40 against 80"""
    from concurrent.futures import ThreadPoolExecutor
    names = sorted(f[len("pt_inst_"):-len(".hip")] for f in os.listdir(HIP) if f.startswith("pt_inst_") and f.endswith(".hip"))
    assert "light_diff" in names and len(names) >= 10, names
    with ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 1)) as pool:
        results = dict(zip(names, pool.map(lambda n: _load_counts(tmp_path, n), names)))
    for name, kernels in results.items():
        assert len(kernels) >= 3, (name, kernels)               # the wave-count builds (x medium off / on where the set has both)
        for k, (scalar, vector) in kernels.items():
            assert scalar >= 60 and vector <= 70 and scalar > vector, (name, k, scalar, vector)
    for k, (scalar, vector) in results["light_diff"].items():
        assert scalar >= 70 and vector <= 55 and scalar > 1.5 * vector, (k, scalar, vector)
