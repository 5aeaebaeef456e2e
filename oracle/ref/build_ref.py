#!/usr/bin/env python3
"""oracle/ref/build_ref.py -- TEST INFRASTRUCTURE: recipe that builds the REFERENCE's own code
for this path from the sources where they lie under /root/reference.  Outputs go only to
oracle/_ref/ (git-ignored binaries) -- no reference text is written into the repository; the
scene-specialised kernel text lives in a temp directory for the duration of the compile.

What gets built
  oracle/_ref/ref_host            reference host code: include/Scene/scene.h (JSON loader),
                                  include/CL/cl_kernel.h (kernel specialiser), Camera
                                  (+ oracle/ref/ref_host.cpp, a 90-line driver)
  (--math libm: the same with the scalar transcendental built-ins taken from glibc's libm instead of prt_detmath.h)
  oracle/_ref/libref_<variant>.so reference device code: kernels/main.cl and everything it
                                  #FILE-includes, specialised for one scene by the reference's
                                  own cl_kernel.h::parse, compiled as OpenCL C for x86-64 by
                                  ROCm clang, linked with oracle/ref/clrt_shim.cpp (OpenCL C
                                  runtime: work-item/image functions + built-in math =
                                  include/prt_detmath.h) and oracle/ref/ref_harness.cpp.
  <blob_dir>/<variant>.sceneblob  the host buffers the reference loader produced (fixture data)

Known, documented interventions:
  Q1  '#FILE:bxdf/materials/..' is lower-case, the directory is 'Materials': resolved through
      a symlink farm in the temp dir (the reference only runs on case-insensitive file systems).
  Q9  the any-hit traversal stack has 8 entries and no overflow check (kernels/geometry/bvh.cl:
      42-44,97): `--shadow-stack N` rewrites that one `#define STACK_SIZE 8` in the TEMP text so
      the host run cannot smash its stack.  Results are identical whenever the original would
      not have overflowed.
  PICK_RANDOM_LIGHT (kernels/integrators/base.cl:9) is a source-level switch whose code reads one element past LIGHT_INDICES:
      `--pick-random-light` flips it in the TEMP text and declares the array one (zero-initialised) element longer.
  C99 `inline` functions without an external definition (kernels/bxdf/Fresnel.cl:33) are
      compiled with -Dinline= so they get one.
"""
import argparse
import os
import shutil
import subprocess
import sys
import tempfile

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
OUT = os.path.join(ROOT, "oracle", "_ref")
CLANG = "/opt/rocm/lib/llvm/bin/clang"
CLANGXX = "/opt/rocm/lib/llvm/bin/clang++"
MARCH = "-march=x86-64-v3"


def run(cmd, **kw):
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, **kw)
    if r.returncode != 0:
        sys.stderr.write("FAILED: %s\n%s\n" % (" ".join(cmd), r.stdout[-4000:]))
        raise SystemExit(1)
    return r.stdout


def make_farm(tmp):
    """symlink farm: <tmp>/kernels mirrors /root/reference/kernels, plus lower-case aliases."""
    kroot = os.path.join(tmp, "kernels")
    src = os.path.join(REF, "kernels")
    for d, dirs, files in os.walk(src):
        rel = os.path.relpath(d, src)
        os.makedirs(os.path.join(kroot, rel), exist_ok=True)
        for f in files:
            os.symlink(os.path.join(d, f), os.path.join(kroot, rel, f))
    for d, dirs, files in os.walk(kroot):
        for sub in list(dirs):
            low = sub.lower()
            if low != sub and not os.path.exists(os.path.join(d, low)):
                os.symlink(os.path.join(d, sub), os.path.join(d, low))
    os.makedirs(os.path.join(tmp, "run"), exist_ok=True)


def set_phase(tmp, phase):
    """The reference selects the phase function by editing the '#FILE:phasefunctions/Isotropic.cl'
    line of kernels/media.cl:61.  Same effect without touching text: point the
    farm's Isotropic.cl link at the requested file."""
    link = os.path.join(tmp, "kernels", "phasefunctions", "Isotropic.cl")
    target = {"": "Isotropic.cl", "isotropic": "Isotropic.cl", "hg": "HenyeyGreenstein.cl", "rayleigh": "Rayleigh.cl"}[phase]
    os.remove(link)
    os.symlink(os.path.join(REF, "kernels", "phasefunctions", target), link)


def build_host():
    os.makedirs(OUT, exist_ok=True)
    exe = os.path.join(OUT, "ref_host")
    run(["g++", "-std=c++17", "-O1", "-w", "-I%s/include" % REF, "-I%s/external" % REF,
         "-I/opt/rocm/include", os.path.join(HERE, "ref_host.cpp"),
         os.path.join(REF, "src/Camera/camera.cpp"), "-o", exe, "-lOpenCL"])
    return exe


def build_support(tmp, math="detmath"):
    objs = []
    for src in ("clrt_shim.cpp", "ref_harness.cpp"):
        o = os.path.join(tmp, src + ".o")
        run([CLANGXX, "-std=c++17", "-O2", "-ffp-contract=off", "-Wall", "-Werror", MARCH, "-fPIC", "-I%s/include" % ROOT] +
            (["-DSHIM_LIBM"] if math == "libm" else []) + ["-c", os.path.join(HERE, src), "-o", o])
        objs.append(o)
    return objs


def build_variant(name, scene, width, height, alpha, shadow_stack, phase, blob_dir, exe, tmp, support, kat=False, view="", pick=False):
    cl = os.path.join(tmp, name + ".cl")
    blob = os.path.join(blob_dir, name + ".sceneblob")
    run([exe, os.path.abspath(scene), str(width), str(height), "1" if alpha else "0", cl, blob],
        cwd=os.path.join(tmp, "run"))
    if shadow_stack or kat or view or pick:
        text = open(cl).read()
        if pick:    # PICK_RANDOM_LIGHT is a source-level switch (kernels/integrators/base.cl:9).  Switched on it indexes LIGHT_INDICES one
                    # past its end with probability 1 / (LIGHT_COUNT + 1) (base.cl:90,204): the array gets one more, zero-initialised,
                    # element in the temp text, so that what is read there is defined (include/prt.h prt_config::pick_random_light)
            for needle, repl in (("#define PICK_RANDOM_LIGHT 0", "#define PICK_RANDOM_LIGHT 1"),
                                 ("__constant uint LIGHT_INDICES[LIGHT_COUNT] = {", "__constant uint LIGHT_INDICES[LIGHT_COUNT + 1] = {")):
                assert text.count(needle) >= 1, "%r not found" % needle     # (header.cl is spliced in more than once, behind its include guard)
                text = text.replace(needle, repl)
        if view:    # the debug views are a source-level switch of kernels/main.cl:15 (VIEW_OPTION ... & VIEW_RESULTS): flip it in the temp text
            needle = "#define VIEW_OPTION (~(0xFF<<(DEBUG*8)) & VIEW_RESULTS)"
            assert text.count(needle) == 1, "VIEW_OPTION define not found exactly once"
            text = text.replace(needle, "#define VIEW_OPTION (~(0xFF<<(DEBUG*8)) & %s)" % view)
        if shadow_stack:
            needle = "#define STACK_SIZE 8\n"
            assert text.count(needle) == 1, "shadow stack define not found exactly once"
            text = text.replace(needle, "#define STACK_SIZE %d\n" % shadow_stack)
        if kat:     # the per-function harness (own code) joins the reference's translation unit, in the temp text only
            text += "\n" + open(os.path.join(HERE, "kat_harness.cl")).read()
        open(cl, "w").write(text)
    obj = os.path.join(tmp, name + ".o")
    run([CLANG, "-x", "cl", "-cl-std=CL1.2", "-Xclang", "-finclude-default-header",
         "-target", "x86_64-unknown-linux-gnu", MARCH, "-O2", "-ffp-contract=off", "-fPIC",
         "-Wno-error=incompatible-pointer-types", "-w", "-Dinline=", "-c", cl, "-o", obj])
    os.remove(cl)
    so = os.path.join(OUT, "libref_%s.so" % name)
    run([CLANGXX, "-shared", "-o", so, obj] + support + ["-Wl,--no-undefined", "-lpthread", "-lm"])
    return so


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--variant", action="append", default=[],
                    help="name=scene.json (repeatable)")
    ap.add_argument("--width", type=int, default=64)
    ap.add_argument("--height", type=int, default=64)
    ap.add_argument("--alpha", action="store_true")
    ap.add_argument("--shadow-stack", type=int, default=64)
    ap.add_argument("--phase", default="", choices=["", "isotropic", "hg", "rayleigh"], help="phase function of the global medium")
    ap.add_argument("--blob-dir", default=os.path.join(ROOT, "tests", "golden"))
    ap.add_argument("--view", default="", choices=["", "VIEW_NORMAL", "VIEW_ALBEDO", "VIEW_SPECULAR", "VIEW_BVH_HIT"],
                    help="debug view of kernels/main.cl:6-15 (VIEW_STACK_INDEX does not compile: Ray has no bvh_stackIndex)")
    ap.add_argument("--math", default="detmath", choices=["detmath", "libm"],
                    help="the OpenCL built-in library behind the kernel text: include/prt_detmath.h (the stated one) or, for the scalar "
                         "transcendentals, the GNU C library's libm (the independent flavour, tools/independent_math.py)")
    ap.add_argument("--pick-random-light", action="store_true", help="PICK_RANDOM_LIGHT 1 (kernels/integrators/base.cl:9) with a defined entry behind LIGHT_INDICES")
    ap.add_argument("--kat", action="store_true", help="append oracle/ref/kat_harness.cl (per-function known-answer entry point kat_run)")
    a = ap.parse_args()
    if not os.path.isdir(REF):
        print("reference not present: nothing to build (the GPU box uses the prebuilt files)")
        return 0
    exe = build_host()
    tmp = tempfile.mkdtemp(prefix="prt_ref_")
    try:
        make_farm(tmp)
        support = build_support(tmp, a.math)
        set_phase(tmp, a.phase)
        for v in a.variant:
            name, scene = v.split("=", 1)
            so = build_variant(name, scene, a.width, a.height, a.alpha, a.shadow_stack, a.phase,
                               a.blob_dir, exe, tmp, support, kat=a.kat, view=a.view, pick=a.pick_random_light)
            print("built", so)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    return 0


if __name__ == "__main__":
    sys.exit(main())
