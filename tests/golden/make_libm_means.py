#!/usr/bin/env python3
"""Generates tests/golden/libm_means.npz from the glibc-libm flavour of the REFERENCE build (oracle/_ref, development container only;
oracle/ref/build_ref.py --math libm: the reference's own kernel text with the GNU C library behind its scalar transcendental built-ins
instead of include/prt_detmath.h).

What it pins: the HIP path against math this repository did NOT write.  A path tracer is chaotic -- the two libraries part ways pixel by
pixel at the first decision an ulp flips -- so what is comparable is the ESTIMATE: BASELINE configs 2 / 3a / 4 at 128 x 128 pixels x 4 096
frames (67 M segments each), stored as the means of the 8 x 8 pixel blocks per channel over their finite pixels (blocks decorrelate the
pixels that share RNG streams).
tests/test_gpu_parity.py::test_hip_agrees_with_the_reference_under_glibc_math renders the same inputs on the GPU and demands the image
means within the Monte-Carlo error of the difference.

usage: python tests/golden/make_libm_means.py      (after __graft_entry__.build() built the *_libm variants)
"""
import importlib
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tests"))

import oracle_api as O  # noqa: E402
from conftest import PKG_NAME  # noqa: E402

SIZE, FRAMES = 128, 4096
# (key, libm build of the reference, scene, phase function (prt_config), environment map)
CASES = [("config2", "cornell_diffuse_libm", "cornell_diffuse.json", 0, False),
         ("config3a", "cornell_roughcond_libm", "cornell_roughcond.json", 0, True),
         ("config4", "cornell_media_hg_libm", "cornell_media.json", 1, True)]


def block_means(img):
    """(SIZE/8, SIZE/8, 3) float64: per 8 x 8 block the mean over its FINITE pixels, the number of those per block, and the number of pixels
    of the image that are not finite (the reference leaves a NaN in a fifth of the pixels of the medium scene: 0 * inf in a weight; such a pixel
    stays NaN for the rest of the render, and which pixels they are differs between two math libraries like everything else)"""
    a = img[..., :3].astype(np.float64)
    ok = np.isfinite(a).all(axis=2)
    a = np.where(ok[..., None], a, 0.0)
    h, w = a.shape[0] // 8, a.shape[1] // 8
    cnt = ok.reshape(h, 8, w, 8).sum(axis=(1, 3))
    sums = a.reshape(h, 8, w, 8, 3).sum(axis=(1, 3))
    return sums / np.maximum(cnt, 1)[..., None], cnt, int((~ok).sum())


def main():
    prt = importlib.import_module(PKG_NAME)
    out = {"size": SIZE, "frames": FRAMES}
    for key, build, scene_json, phase, use_env in CASES:
        scene = prt.HostScene(scene_json)
        cam = prt.default_camera(SIZE, SIZE)
        seeds = prt.seed_pairs(FRAMES)
        env = prt.make_sky(64, 32) if use_env else None
        state, img = O.RefOracle(build).render(scene.desc, bytes(cam), SIZE, SIZE, seeds, env=env, threads=8)
        blocks, cnt, bad = block_means(img.reshape(SIZE, SIZE, 4))
        out[key + "_blocks"] = blocks
        out[key + "_counts"] = cnt.astype(np.int32)
        out[key + "_nonfinite"] = bad
        out[key + "_segments"] = float(state["acc"][:, 3].sum())
        print(key, "means", (blocks * cnt[..., None]).sum(axis=(0, 1)) / cnt.sum(), "non-finite pixels", bad, "segments", out[key + "_segments"])
    np.savez_compressed(os.path.join(HERE, "libm_means.npz"), **out)


if __name__ == "__main__":
    main()
