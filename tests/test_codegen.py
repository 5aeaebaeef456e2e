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


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="no hipcc")
def test_uniform_loads_of_the_render_kernel_stay_scalar(tmp_path):
    out = tmp_path / "light_diff.s"
    cmd = [HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-fast-math", "-fno-slp-vectorize", "-S",
           "--cuda-device-only", "-I" + os.path.join(ROOT, "include"), "-I" + HIP, "-o", str(out), os.path.join(HIP, "pt_inst_light_diff.hip")]
    subprocess.run(cmd, check=True, capture_output=True, timeout=900)
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
    assert len(kernels) >= 4, kernels                      # medium off / on x the wave-count builds
    for name, (scalar, vector) in kernels.items():
        # shipped: 82 ... 110 scalar against 38 ... 44 vector loads; de-scalarised: about 40 against 80
        assert scalar >= 70 and vector <= 55 and scalar > 1.5 * vector, (name, scalar, vector)
