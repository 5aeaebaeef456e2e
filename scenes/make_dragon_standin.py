#!/usr/bin/env python3
"""Procedural stand-in for resources/models/dragon.obj (~871 k triangles), which is listed in the
reference's .MISSING_LARGE_BLOBS and is not available: a bumpy (2,3) torus-knot tube, tessellated
1320 x 330, with smooth per-vertex normals, scaled to sit inside the Cornell box.  Deterministic
(pure closed-form geometry, float64 math rounded once to float32).  Writes the ".prtmesh" soup
(csrc/host/model_loader.h).  62 MB: generated on demand, never committed.

usage: python scenes/make_dragon_standin.py [out.prtmesh] [n_u n_v]
"""
import os
import struct
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
DEFAULT_OUT = os.path.join(HERE, "models", "dragon_standin.prtmesh")


def generate(n_u=1320, n_v=330):
    u = np.linspace(0.0, 2.0 * np.pi, n_u, endpoint=False)[:, None]
    v = np.linspace(0.0, 2.0 * np.pi, n_v, endpoint=False)[None, :]
    p, q = 2.0, 3.0
    # centre curve of the knot and its Frenet-like frame
    r = 0.75 + 0.3 * np.cos(q * u)
    c = np.concatenate([r * np.cos(p * u), 0.45 * np.sin(q * u), r * np.sin(p * u)], axis=1)      # (n_u, 3)
    du = 2.0 * np.pi / n_u
    t = (np.roll(c, -1, axis=0) - np.roll(c, 1, axis=0)) / (2 * du)
    t /= np.linalg.norm(t, axis=1, keepdims=True)
    up = np.array([0.0, 1.0, 0.0])
    b = np.cross(t, up)
    b /= np.linalg.norm(b, axis=1, keepdims=True)
    n = np.cross(b, t)
    bump = 1.0 + 0.18 * np.sin(9.0 * v) * np.sin(40.0 * u) + 0.08 * np.sin(23.0 * u + 5.0 * v)
    tube = 0.17 * bump                                                                             # (n_u, n_v)
    pos = c[:, None, :] + tube[..., None] * (np.cos(v)[..., None] * n[:, None, :] + np.sin(v)[..., None] * b[:, None, :])
    # smooth normals from the parametric derivatives (periodic central differences)
    dpu = np.roll(pos, -1, axis=0) - np.roll(pos, 1, axis=0)
    dpv = np.roll(pos, -1, axis=1) - np.roll(pos, 1, axis=1)
    nor = np.cross(dpv, dpu)
    nor /= np.linalg.norm(nor, axis=2, keepdims=True)
    # place in the box: x,z in [-1.3, 1.3], resting a little above the floor y = 0
    pos *= 1.05
    pos[..., 1] += 1.15 - 0.5 * (pos[..., 1].max() + pos[..., 1].min())
    i0 = np.arange(n_u)[:, None]
    j0 = np.arange(n_v)[None, :]
    i1, j1 = (i0 + 1) % n_u, (j0 + 1) % n_v
    vert = np.concatenate([pos, nor], axis=2).astype(np.float32)                                   # (n_u, n_v, 6)
    a, b_, c_, d = vert[i0, j0], vert[i1, j0], vert[i1, j1], vert[i0, j1]
    tris = np.stack([np.stack([a, b_, c_], axis=2), np.stack([a, c_, d], axis=2)], axis=2)         # (n_u, n_v, 2, 3, 6)
    return np.ascontiguousarray(tris.reshape(-1, 3, 6))


def write(path=DEFAULT_OUT, n_u=1320, n_v=330):
    tris = generate(n_u, n_v)
    os.makedirs(os.path.dirname(path), exist_ok=True)
    with open(path, "wb") as f:
        f.write(b"PRTMESH1")
        f.write(struct.pack("<I", tris.shape[0]))
        f.write(tris.tobytes())
    return tris.shape[0]


if __name__ == "__main__":
    out = sys.argv[1] if len(sys.argv) > 1 else DEFAULT_OUT
    nu, nv = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (1320, 330)
    print(write(out, nu, nv), "triangles ->", out)
