#!/usr/bin/env python3
"""Builds libprt.so in-tree: hipcc --offload-arch=gfx950 for the kernels + C-ABI, host C++ for the
scene/camera/OBJ/BVH model.  -ffp-contract=off everywhere: the arithmetic contract of
include/prt_detmath.h forbids implicit fma contraction on both host and device."""
import hashlib
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "build")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
ARCH = "gfx950"
HOSTCXX = os.environ.get("CXX", "g++")

COMMON = ["-std=c++17", "-O3", "-fPIC", "-ffp-contract=off", "-fno-fast-math", "-Wall",
          "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(CSRC, "host"), "-I" + os.path.join(CSRC, "hip")]

HOST_SOURCES = ["host/scene.cpp", "host/camera.cpp", "host/model_loader.cpp", "host/bvh.cpp", "host/hdr_loader.cpp", "host/host_capi.cpp"]
# the render kernel is instantiated per compile-time material set in a file of its own (pt_inst_*.hip): they compile in parallel
HIP_SOURCES = ["hip/prt_api.cpp", "hip/pt_pack.cpp", "hip/pt_kernels.hip", "hip/prt_build_id.cpp"] + \
    ["hip/pt_inst_%s.hip" % k for k in ("light_diff", "coat", "rough_cond", "rough_diel", "generic", "sdf", "view", "view_sdf", "pick", "envis")]


HIP_FLAGS = ["-fno-slp-vectorize", "-x", "hip", "--offload-arch=" + ARCH]


def source_files(root=None):
    """every file libprt.so is made of: the sources and headers under csrc/, include/*.h and this recipe (the flags live here)"""
    root = root or ROOT
    csrc = os.path.join(root, os.path.basename(HERE), "csrc")
    files = []
    for d, _, names in os.walk(csrc):
        files += [os.path.join(d, f) for f in names if f.endswith((".h", ".cpp", ".hip"))]
    files += [os.path.join(root, "include", f) for f in os.listdir(os.path.join(root, "include")) if f.endswith(".h")]
    files.append(os.path.join(root, os.path.basename(HERE), "build.py"))
    return sorted(files)


def source_build_id(root=None, extra=()):
    """what prt_build_id() of a library built from the working tree returns: a hash of the CONTENT of every source, header and flag
    (paths relative to the repository, so the id is the same in any checkout and on the GPU box)"""
    root = root or ROOT
    h = hashlib.sha256()
    for f in source_files(root):
        h.update(os.path.relpath(f, root).encode() + b"\0")
        with open(f, "rb") as fh:
            h.update(fh.read())
        h.update(b"\0")
    for x in list(COMMON[:6]) + HIP_FLAGS + list(extra):        # (the -I paths are the checkout's: not part of the id)
        h.update(x.encode() + b"\0")
    return h.hexdigest()[:16]


def newer(src, obj, deps):
    if not os.path.exists(obj):
        return True
    t = os.path.getmtime(obj)
    return any(os.path.getmtime(d) > t for d in [src] + deps)


def run(cmd):
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        sys.stderr.write("FAILED: %s\n%s\n" % (" ".join(cmd), r.stdout[-8000:]))
        raise SystemExit(1)
    return r.stdout


def build(verbose=False, extra_hip_flags=()):
    os.makedirs(OBJ, exist_ok=True)
    headers = []
    for d, _, files in os.walk(CSRC):
        headers += [os.path.join(d, f) for f in files if f.endswith(".h")]
    headers += [os.path.join(ROOT, "include", f) for f in os.listdir(os.path.join(ROOT, "include"))]
    headers.append(os.path.abspath(__file__))           # the flags live here
    objs = []
    jobs = []
    # the library says what it was built from (prt_build_id(), include/prt.h): the id is compiled into prt_build_id.cpp, which is
    # recompiled whenever the id on record differs from the working tree's
    build_id = source_build_id(extra=extra_hip_flags)
    id_file = os.path.join(OBJ, "build_id.txt")
    id_stale = (not os.path.exists(id_file)) or open(id_file).read().strip() != build_id
    for s in HOST_SOURCES + HIP_SOURCES:
        src = os.path.join(CSRC, s)
        obj = os.path.join(OBJ, s.replace("/", "_") + ".o")
        objs.append(obj)
        is_id = s == "hip/prt_build_id.cpp"
        if not newer(src, obj, headers) and not (is_id and id_stale):
            continue
        if s in HIP_SOURCES:
            # -fno-slp-vectorize: left on, the SLP vectorizer turns pairs of scalar float operations of the kernels into v_pk_*_f32
            # with the register shuffling that takes -- same bits, 4 % slower
            cmd = [HIPCC] + COMMON + HIP_FLAGS + list(extra_hip_flags)
            if is_id:
                cmd.append('-DPRT_BUILD_ID="%s"' % build_id)
        else:
            cmd = [HOSTCXX] + COMMON
        jobs.append(cmd + ["-c", src, "-o", obj])
    if jobs:
        from concurrent.futures import ThreadPoolExecutor
        with ThreadPoolExecutor(max_workers=max(1, min(len(jobs), int(os.environ.get("PRT_BUILD_JOBS", "0")) or (os.cpu_count() or 4)))) as pool:
            for out in pool.map(run, jobs):
                if verbose and out.strip():
                    print(out)
    lib = os.path.join(HERE, "libprt.so")
    if (not os.path.exists(lib)) or any(os.path.getmtime(o) > os.path.getmtime(lib) for o in objs):
        run([HIPCC, "-shared", "-fPIC", "--offload-arch=" + ARCH, "-o", lib] + objs)
    with open(id_file, "w") as fh:
        fh.write(build_id + "\n")
    exe = os.path.join(HERE, "prt_render")
    src = os.path.join(CSRC, "host", "prt_render.cpp")
    if newer(src, exe, headers + [lib]):
        run([HOSTCXX] + COMMON + [src, "-o", exe, "-L" + HERE, "-lprt", "-Wl,-rpath,$ORIGIN"])
    return lib


if __name__ == "__main__":
    flags = [a for a in sys.argv[1:] if a.startswith("-")]
    print(build(verbose=True, extra_hip_flags=flags))
