#!/usr/bin/env python3
"""tools/oracle_coverage.py -- which branches of the reference algorithm do the committed fixtures execute?

Builds oracle/pt_oracle.c (the line-by-line cited restatement, bit-identical to the reference build on every fixture)
with gcov instrumentation, replays exactly the inputs of the golden fixtures (tests/golden/*.npz: same scenes, cameras,
sizes, frame counts, seeds -- and checks it reproduces them), and prints the source lines of pt_oracle.c that were never
executed together with the function they belong to (every function cites the reference file:line it restates).
The committed report is profiles/r04_oracle_coverage.txt (round 2's: r02_).
usage: python tools/oracle_coverage.py > profiles/r04_oracle_coverage.txt"""
import ctypes as C
import importlib
import os
import re
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
WORK = "/tmp/prt_cov"

# why a line / outcome stays unreached: (substring of the source line or function name) -> reason
WHY = [
    ("checkRefractionConstraint", "only called by DielectricBSDF_eval/_pdf, see there"),
    ("ConductorBSDF_eval", "unreachable in the reference: BSDF_eval2 / BSDF_pdf run only in lightSample, which handleSurface calls only for materials with a non-specular lobe (base.cl:168); a mirror has none and the JSON cannot set lobes"),
    ("ConductorBSDF_pdf", "as ConductorBSDF_eval"),
    ("DielectricBSDF_eval", "as ConductorBSDF_eval: a smooth dielectric has only specular lobes"),
    ("DielectricBSDF_pdf", "as ConductorBSDF_eval"),
    ("PRT_MAT_COND) f =", "as ConductorBSDF_eval"), ("PRT_MAT_DIEL) f =", "as ConductorBSDF_eval"),
    ("PRT_MAT_COND) return", "as ConductorBSDF_eval"), ("PRT_MAT_DIEL) return", "as ConductorBSDF_eval"),
    ("if (F == 1.0f) return 0", "unreachable: the branch is the else of `next1D() < F` and next1D() < 1 (Dielectric.cl:19-33)"),
    ("LambertBSDF]", "wi.z <= 0 needs wi.z == 0 exactly: intersect_scene flips the shading normal towards the ray for non-transmissive materials (intersect.cl:229-233)"),
    ("LambertBSDF_eval", "the wi.z <= 0 half: as LambertBSDF (the wo.z <= 0 half is taken)"),
    ("RoughConductorBSDF_pdf", "unreachable: BSDF_pdf is evaluated only when BSDF_eval2 != 0 (base.cl:112-127), which already needs wi.z > 0 and wo.z > 0"),
    ("CoatBSDF_pdf", "the zero exit: as RoughConductorBSDF_pdf; the reflection-constraint exit needs the light direction to be the exact mirror direction"),
    ("pm < 1e-10f", "needs a microfacet pdf below 1e-10 (RoughDielectric.cl:33): not met by any fixture"),
    ("Microfacet_", "dist is 1 << n from the JSON (scene.h:90-93): one of Beckmann / Phong / GGX always matches"),
    ("s_map", "an SDF mesh always carries one of the four SDF type bits (scene.h)"),
    ("sampleDirect", "the light is a sphere or a quad in every scene the reference can build (an SDF light cannot be sampled, geometry.cl:11-32)"),
    ("directPdf", "as sampleDirect"),
    ("geom_flags & PRT_GEOM", "scenes without spheres (or without a light) do not compile in the reference (geometry.cl:17-24, base.cl:168-172): every fixture has spheres and quads; the SDF half is taken by cornell_sdf"),
    ("intersect_sphere(ray, &sc->meshes[i])) { if (ray->t < maxDist)", "an accepted hit always has t < maxDist: the inner test cannot fail"),
    ("intersect_quad(&sc->meshes[fl++], ray)) { if (ray->t < maxDist)", "as the sphere line above"),
    ("g == 0.0f", "g is the source-level constant 0.6 in the reference (HenyeyGreenstein.cl:4)"),
    ("uniformSphere(xi_x, xi_y)", "as g == 0"), ("INV_FOUR_PI;", "as g == 0"),
    ("vdot(f, f) == 0.0f) return vsplat", "a phase function value is never 0"),
    ("!phase_sample", "phase_sample always succeeds"),
    ("ray->t <= EPS", "a closest hit has t > EPS by construction (triangle.cl:36)"),
    ("stackSize < PTO_STACK", "harness guard, not reference code"),
    ("active_mats & PRT_MAT_LIGHT) && (mat.t", "a scene without a light does not compile in the reference"),
    ("active_mats & PRT_MAT_LIGHT) && (mat->lobes", "as above"),
    ("PRT_MAT_COAT) return CoatBSDF", "a material always has one of the compiled-in types"), ("PRT_MAT_COAT) f =", "as above"), ("PRT_MAT_COAT) return CoatBSDF_pdf", "as above"),
    ("[BSDF]", "a material always has one of the compiled-in types"), ("[BSDF_eval2]", "as above"), ("[BSDF_pdf]", "as above"),
    ("!BSDF2(sc, e, ray, mat, rng)) return 1", "the non-MIS branch serves the purely specular materials, whose samplers never fail (Conductor.cl:4-12; Dielectric.cl: see F == 1)"),
    ("pto_render", "harness (argument checks, threading), not reference code"),
]


def why(fn, src):
    for key, reason in WHY:
        if key in src or key == fn or key == "[%s]" % fn or key == fn + "]" or (key.endswith("_") and fn.startswith(key)):
            return reason
    return "NOT EXPLAINED"


def replay(so):
    """renders every golden fixture's input with the instrumented oracle; the counters are written when this process exits"""
    prt = importlib.import_module("photorealistic-rendering-using-opencl_amd")
    import oracle_api as O
    from conftest import GOLDEN, VARIANTS, variant_camera, variant_config
    rs = O.Restatement(so)
    replayed = []
    for name, (scene_json, phase, use_env) in VARIANTS.items():
        gpath = os.path.join(GOLDEN, name + ".npz")
        g = np.load(gpath) if os.path.exists(gpath) else None
        W, H, frames = (int(g["width"]), int(g["height"]), int(g["frames"])) if g is not None else (int(os.environ.get("COV_W", 48)), int(os.environ.get("COV_H", 36)), int(os.environ.get("COV_FRAMES", 256)))
        scene = prt.HostScene(scene_json)
        cfg = variant_config(scene, name)                      # (-alpha, PICK_RANDOM_LIGHT: what the reference build of the fixture was given)
        cfg.phase_function = phase
        cam = variant_camera(prt, name, W, H)
        env = prt.make_sky(64, 32) if use_env else None
        state, img = rs.render(cfg, scene.desc, cam, W, H, prt.seed_pairs(frames), env=env, threads=4)
        if g is not None:
            gstate = np.ascontiguousarray(g["state"]).view(O.PATH_STATE_DTYPE).reshape(-1)
            assert not O.state_fields_equal(gstate, state) and O.images_equal(g["image"], img), name
        replayed.append("%s %dx%dx%d%s" % (name, W, H, frames, "" if g is not None else " (no golden yet)"))
    from conftest import VIEW_VARIANTS
    for fixture, (base, view) in VIEW_VARIANTS.items():      # the debug views of main.cl:6-15 (prt_config::view_option)
        g = np.load(os.path.join(GOLDEN, fixture + ".npz"))
        scene_json, phase, use_env = VARIANTS[base]
        W, H, frames = int(g["width"]), int(g["height"]), int(g["frames"])
        scene = prt.HostScene(scene_json)
        cfg = variant_config(scene, base)
        cfg.phase_function = phase
        cfg.view_option = view
        state, img = rs.render(cfg, scene.desc, variant_camera(prt, base, W, H), W, H, prt.seed_pairs(frames), env=prt.make_sky(64, 32) if use_env else None, threads=4)
        gstate = np.ascontiguousarray(g["state"]).view(O.PATH_STATE_DTYPE).reshape(-1)
        assert not O.state_fields_equal(gstate, state) and O.images_equal(g["image"], img), fixture
        replayed.append("%s %dx%dx%d" % (fixture, W, H, frames))
    g = np.load(os.path.join(GOLDEN, "cornell_diffuse_spp.npz"))
    scene = prt.HostScene("cornell_diffuse.json")
    rs.render(scene.config(), scene.desc, prt.default_camera(int(g["width"]), int(g["height"])), int(g["width"]), int(g["height"]),
              prt.seed_pairs(int(g["frames"])), spp_limit=int(g["spp"]), threads=4)
    replayed.append("cornell_diffuse_spp")
    print(", ".join(replayed))


def main():
    os.makedirs(WORK, exist_ok=True)
    for f in os.listdir(WORK):
        if f.endswith((".gcda", ".gcno", ".gcov")):
            os.remove(os.path.join(WORK, f))
    so = os.path.join(WORK, "liboracle_cov.so")
    subprocess.check_call(["gcc", "-std=c11", "-O0", "-ffp-contract=off", "-fno-fast-math", "-march=x86-64-v3", "-fPIC", "--coverage",
                           "-fprofile-update=atomic", "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "oracle"),
                           "-shared", "-o", so, os.path.join(ROOT, "oracle", "pt_oracle.c"), "-lpthread", "-lm"], cwd=WORK)
    replayed = subprocess.check_output([sys.executable, os.path.abspath(__file__), "--replay", so], text=True).strip().split("\n")[-1]
    for ext in (".gcda", ".gcno"):        # gcc names them after the output file when compiling and linking in one step
        os.replace(os.path.join(WORK, "liboracle_cov.so-pt_oracle" + ext), os.path.join(WORK, "pt_oracle" + ext))
    subprocess.check_call(["gcov", "-b", "-c", "-o", WORK, os.path.join(ROOT, "oracle", "pt_oracle.c")], cwd=WORK, stdout=subprocess.DEVNULL)
    lines = open(os.path.join(WORK, "pt_oracle.c.gcov"), errors="replace").read().split("\n")
    func, never, total, hit = "?", [], 0, 0
    br_total = br_taken = 0
    half = []                       # conditions of which only one outcome was ever taken
    last_src = (0, "")
    cur_br = []
    for ln in lines:
        if ln.startswith("branch"):
            br_total += 1
            taken = "never executed" not in ln and not re.search(r"taken 0\b", ln)
            br_taken += 1 if taken else 0
            cur_br.append(taken)
            continue
        m = re.match(r"\s*([^:]+):\s*(\d+):(.*)", ln)
        if not m:
            continue
        if cur_br and any(cur_br) and not all(cur_br) and last_src[0]:
            half.append((last_src[0], last_src[2], last_src[1]))
        cur_br = []
        cnt, no, src = m.group(1).strip(), int(m.group(2)), m.group(3)
        fm = re.match(r"(?:static\s+)?(?:inline\s+)?[A-Za-z_][A-Za-z0-9_ \*]*?\b([A-Za-z_][A-Za-z0-9_]*)\s*\([^;]*\)\s*\{", src)
        if fm and not src.startswith((" ", "\t")):
            func = fm.group(1)
        last_src = (no, src.strip(), func)
        if cnt == "-":
            continue
        total += 1
        if cnt in ("#####", "====="):
            never.append((no, func, src.strip()))
        else:
            hit += 1
    print("oracle/pt_oracle.c replaying the golden fixtures' inputs (%s):" % replayed)
    print("  %d of %d executable lines run (%.1f %%), %d of %d branch outcomes taken (%.1f %%)"
          % (hit, total, 100.0 * hit / total, br_taken, br_total, 100.0 * br_taken / max(br_total, 1)))
    print("lines never executed:")
    for no, fn, src in never:
        print("  pt_oracle.c:%d  [%s]  %s\n        -> %s" % (no, fn, src[:120], why(fn, src)))
    print("conditions with an outcome never taken (%d):" % len(half))
    for no, fn, src in half:
        print("  pt_oracle.c:%d  [%s]  %s\n        -> %s" % (no, fn, src[:120], why(fn, src)))


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--replay":
        replay(sys.argv[2])
    else:
        main()
