#!/usr/bin/env python3
"""Generates the benchmark / parity scene files of BASELINE.json's configs.

The Cornell-box layout (one sphere light, six quads, one OBJ) is the one the reference's
scenes/cornell.json describes; the files are generated from the parameters below rather than
copied, and the per-config differences are the ones SURVEY §8d lists:
  cornell_coat        config 1: teapot COAT (type 4) roughness 0.1, Beckmann        (= reference cornell.json)
  cornell_diffuse     config 2: teapot DIFF (type 1)  -> ACTIVE_MATS = LIGHT|DIFF
  cornell_roughcond   config 3a: teapot ROUGH_COND (type 10), GGX (dist 2), roughness 0.1
  cornell_roughdiel   config 3b: teapot ROUGH_DIEL (type 11), GGX, roughness 0.1, + one small DIEL sphere
  cornell_media       config 4: global medium density 0.07 sigmaA 0 sigmaS 1, teapot DIFF, light (0,1.5,-1) r 0.2 L 34
  cornell_dragon      config 5: procedural ~870k-triangle stand-in mesh (dragon.obj is not available)
"""
import json
import os

HERE = os.path.dirname(os.path.abspath(__file__))

WHITE = [1.0, 1.0, 1.0]


def quad(base, e0, e1, color):
    return {"vertices": base + e0 + e1, "material": {"color": color}}


def box_quads():
    return [
        quad([0.0, 0.0, 0.0], [4.0, 0.0, 0.0], [0.0, 0.0, 4.0], WHITE),      # ceiling-side plane y=0 facing -y
        quad([0.0, 4.0, 0.0], [-4.0, 0.0, 0.0], [0.0, 0.0, 4.0], WHITE),
        quad([0.0, 2.0, 2.0], [4.0, 0.0, 0.0], [0.0, 4.0, 0.0], WHITE),
        quad([0.0, 2.0, -2.0], [-4.0, 0.0, 0.0], [0.0, 4.0, 0.0], WHITE),
        quad([2.0, 2.0, 0.0], [0.0, 4.0, 0.0], [0.0, 0.0, 4.0], [0.8, 0.1, 0.1]),
        quad([-2.0, 2.0, 0.0], [0.0, -4.0, 0.0], [0.0, 0.0, 4.0], [0.1, 0.8, 0.1]),
    ]


def settings(b=32, d=8, s=32, t=64, sc=128):
    return {"MAX_BOUNCES": b, "MAX_DIFF_BOUNCES": d, "MAX_SPEC_BOUNCES": s, "MAX_TRANS_BOUNCES": t,
            "MAX_SCATTERING_EVENTS": sc, "MARCHING_STEPS": 128, "SHADOW_MARCHING_STEPS": 64}


def cornell(obj_material, obj="teapot.obj", extra_spheres=(), light=None, medium=None, st=None):
    light = light or {"pos": [0.0, 3.0, 0.0], "radius": 0.5, "material": {"color": [5.0, 5.0, 5.0], "type": 0}}
    doc = {}
    if medium:
        doc["global_medium"] = medium
    doc["settings"] = st or settings()
    doc["scene"] = {"obj": {"path": obj, "material": obj_material},
                    "spheres": [light] + list(extra_spheres), "quads": box_quads()}
    return doc


SCENES = {
    "cornell_coat": cornell({"color": WHITE, "type": 4, "roughness": 0.1}),
    "cornell_diffuse": cornell({"color": WHITE, "type": 1}),
    "cornell_roughcond": cornell({"color": WHITE, "type": 10, "dist": 2, "roughness": 0.1}),
    "cornell_roughdiel": cornell({"color": WHITE, "type": 11, "dist": 2, "roughness": 0.1},
                                 extra_spheres=[{"pos": [1.2, 0.4, 0.8], "radius": 0.4,
                                                 "material": {"color": WHITE, "type": 3}}]),
    "cornell_media": cornell({"color": WHITE, "type": 1},
                             light={"pos": [0.0, 1.5, -1.0], "radius": 0.2,
                                    "material": {"color": [34.0, 34.0, 34.0], "type": 0}},
                             medium={"density": 0.07, "sigmaA": 0.0, "sigmaS": 1.0},
                             st=settings(12, 4, 16, 32, 1024)),
    "cornell_dragon": cornell({"color": WHITE, "type": 1}, obj="dragon_standin.prtmesh"),
    # coverage scenes (not BASELINE configs): every material / distribution / light type of SURVEY s8a
    "cornell_mixed": cornell({"color": [0.9, 0.8, 0.6], "type": 10, "dist": 1, "roughness": 0.25},          # ROUGH_COND, Phong
                             extra_spheres=[{"pos": [-1.1, 0.45, 0.6], "radius": 0.45, "material": {"color": [0.95, 0.95, 0.95], "type": 2}},               # COND mirror
                                            {"pos": [1.1, 0.5, 0.9], "radius": 0.5, "material": {"color": [0.6, 0.9, 0.7], "type": 3, "absorptive": 1}},    # DIEL, ABS_REFR
                                            {"pos": [0.2, 0.35, 1.4], "radius": 0.35,
                                             "material": {"color": [0.9, 0.7, 0.9], "type": 11, "dist": 0, "roughness": 0.15, "absorptive": 2}}]),          # ROUGH_DIEL Beckmann, ABS_REFR2
    # SDF primitives ("next" row N4): raymarched sphere, box, round box and a tilted plane next to the teapot
    "cornell_sdf": dict(cornell({"color": WHITE, "type": 1}), **{}),
    "cornell_quadlight": {"settings": settings(16, 6, 16, 16, 16),
                          "scene": {"obj": {"path": "teapot.obj", "material": {"color": WHITE, "type": 4, "dist": 2, "roughness": 0.2}},   # COAT over GGX
                                    "spheres": [{"pos": [1.0, 0.4, -0.8], "radius": 0.4, "material": {"color": [0.7, 0.7, 0.9], "type": 1}}],
                                    "quads": [quad([0.0, 3.95, 0.0], [-1.2, 0.0, 0.0], [0.0, 0.0, 1.2], [12.0, 12.0, 12.0])] + box_quads()}},
}
SCENES["cornell_quadlight"]["scene"]["quads"][0]["material"]["type"] = 0        # the quad is the (only) light
SCENES["cornell_sdf"]["settings"] = dict(settings(12, 6, 12, 12, 12), MARCHING_STEPS=96, SHADOW_MARCHING_STEPS=48)
SCENES["cornell_sdf"]["scene"]["sdfs"] = [
    {"pos": [-1.2, 0.45, 0.9], "type": 4, "params": [0.45], "material": {"color": [0.9, 0.6, 0.3], "type": 1}},                    # SDF_SPHERE
    {"pos": [1.25, 0.35, 0.7], "type": 5, "params": [0.35, 0.35, 0.35], "material": {"color": [0.3, 0.6, 0.9], "type": 4, "roughness": 0.2}},   # SDF_BOX, COAT
    {"pos": [0.9, 0.3, -1.0], "type": 6, "params": [0.3, 0.2, 0.3, 0.08], "material": {"color": [0.8, 0.8, 0.8], "type": 10, "dist": 2, "roughness": 0.3}},  # SDF_ROUND_BOX
    {"pos": [0.0, 0.0, -1.9], "type": 7, "params": [0.0, 0.2425356, 0.9701425, 0.0], "material": {"color": [0.6, 0.6, 0.6], "type": 1}},   # SDF_PLANE
]

# branch-coverage scenes (round 2, profiles/r02_oracle_coverage.txt): the rarely taken branches of the reference
# tiny mesh = a BVH whose root is a leaf (bvh.cl:34-40,121-128) with skewed vertex normals (shading normal against the
# geometric side: the wi.z <= 0 exits of the BSDFs), low bounce caps (pathtracing.cl:109-115), a large clear glass
# sphere in front of the camera (total internal reflection, Fresnel.cl:47), a mirror, a nearly smooth rough dielectric
SCENES["cornell_edge"] = cornell({"color": [0.9, 0.9, 0.8], "type": 4, "roughness": 0.05}, obj="tiny.obj",
                                 extra_spheres=[{"pos": [0.3, 1.1, 1.4], "radius": 0.9, "material": {"color": WHITE, "type": 3}},
                                                {"pos": [-1.3, 0.5, 0.2], "radius": 0.5, "material": {"color": [0.9, 0.9, 0.9], "type": 2}},
                                                {"pos": [1.3, 0.45, -0.3], "radius": 0.45, "material": {"color": WHITE, "type": 11, "dist": 2, "roughness": 0.001}},
                                                {"pos": [-0.6, 0.3, 1.5], "radius": 0.3, "material": {"color": [0.7, 0.8, 0.9], "type": 1}}],
                                 st=settings(6, 3, 2, 3, 4))
# an absorbing-only medium (media/homogeneous.cl:16-21) and a dense one that runs into MAX_SCATTERING_EVENTS (pathtracing.cl:38)
# (the two-triangle mesh again: as clear glass it is hit from behind at grazing angles -> total internal reflection,
#  Fresnel.cl:47 and the F == 1 exit of Dielectric.cl; as Lambert its skewed normals give wi.z <= 0, Lambert.cl:6,19)
SCENES["cornell_absfog"] = cornell({"color": WHITE, "type": 3}, obj="tiny.obj", medium={"density": 0.5, "sigmaA": 0.6, "sigmaS": 0.0}, st=settings(12, 4, 16, 32, 8))
SCENES["cornell_fogcap"] = cornell({"color": WHITE, "type": 1}, obj="tiny.obj",
                                   light={"pos": [0.0, 1.5, -1.0], "radius": 0.2, "material": {"color": [34.0, 34.0, 34.0], "type": 0}},
                                   medium={"density": 1.5, "sigmaA": 0.05, "sigmaS": 1.0}, st=settings(24, 6, 16, 32, 2))

# every material type + a global medium compiled in at once: the reference build the per-function known-answer vectors
# come from (tests/golden/make_kat.py; the harness passes its own Material / Mesh / medium records at run time)
SCENES["kat_all"] = cornell({"color": WHITE, "type": 4, "roughness": 0.1},
                            extra_spheres=[{"pos": [-1.2, 0.4, 0.5], "radius": 0.4, "material": {"color": WHITE, "type": 1}},
                                           {"pos": [-0.4, 0.4, 1.2], "radius": 0.4, "material": {"color": WHITE, "type": 2}},
                                           {"pos": [0.4, 0.4, 1.2], "radius": 0.4, "material": {"color": WHITE, "type": 3}},
                                           {"pos": [1.2, 0.4, 0.5], "radius": 0.4, "material": {"color": WHITE, "type": 10, "dist": 2, "roughness": 0.2}},
                                           {"pos": [1.2, 1.3, 0.5], "radius": 0.4, "material": {"color": WHITE, "type": 11, "dist": 1, "roughness": 0.2}}],
                            medium={"density": 0.07, "sigmaA": 0.1, "sigmaS": 1.0})

# two lights -- a sphere and a quad -- behind a non-emitting sphere at mesh index 0: PICK_RANDOM_LIGHT (kernels/integrators/base.cl:9,88-93;
# prt_config::pick_random_light) picks LIGHT_INDICES[0], LIGHT_INDICES[1] or the entry behind the array (defined as 0: the diffuse sphere
# is then sampled as if it were a light); the _fog scene runs the same choice in volumeLightSample (base.cl:202-207)
def _twolights(medium=None, st=None):
    doc = cornell({"color": WHITE, "type": 1}, medium=medium, st=st,
                  extra_spheres=[])
    doc["scene"]["spheres"] = [{"pos": [-1.1, 0.45, 0.7], "radius": 0.45, "material": {"color": [0.7, 0.7, 0.9], "type": 1}},
                               {"pos": [0.6, 3.0, 0.0], "radius": 0.4, "material": {"color": [6.0, 5.0, 4.0], "type": 0}}]
    doc["scene"]["quads"] = [quad([-1.2, 3.95, 0.6], [-1.0, 0.0, 0.0], [0.0, 0.0, 1.0], [3.0, 6.0, 9.0])] + box_quads()
    doc["scene"]["quads"][0]["material"]["type"] = 0
    return doc


SCENES["cornell_twolights"] = _twolights(st=settings(16, 6, 16, 16, 16))
SCENES["cornell_twolights_fog"] = _twolights(medium={"density": 0.25, "sigmaA": 0.05, "sigmaS": 1.0}, st=settings(12, 4, 16, 32, 64))

# an OPEN box (floor, back wall, one side wall) under the sky: the scene of the environment-map importance-sampling tests
# (prt_config::env_importance_sampling; in the closed Cornell box the map is only ever seen by camera rays)
SCENES["cornell_open"] = cornell({"color": WHITE, "type": 1}, st=settings(12, 6, 12, 12, 12),
                                 extra_spheres=[{"pos": [1.1, 0.45, 0.6], "radius": 0.45, "material": {"color": [0.9, 0.8, 0.6], "type": 10, "dist": 2, "roughness": 0.3}},
                                                {"pos": [-1.2, 0.4, 0.9], "radius": 0.4, "material": {"color": [0.95, 0.95, 0.95], "type": 4, "roughness": 0.15}}],
                                 light={"pos": [0.0, 3.0, 0.0], "radius": 0.25, "material": {"color": [8.0, 8.0, 8.0], "type": 0}})
SCENES["cornell_open"]["scene"]["quads"] = [box_quads()[k] for k in (0, 3, 5)]

if __name__ == "__main__":
    for name, doc in SCENES.items():
        with open(os.path.join(HERE, name + ".json"), "w") as f:
            json.dump(doc, f, separators=(",", ":"))
            f.write("\n")
        print("wrote", name)
