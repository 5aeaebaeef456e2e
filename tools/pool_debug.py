#!/usr/bin/env python3
"""development helper: the pool kernel against render_kernel on one golden's inputs; says where the two differ"""
import importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
prt = importlib.import_module("photorealistic-rendering-using-opencl_amd")
variant = sys.argv[1] if len(sys.argv) > 1 else "cornell_diffuse"
W = int(sys.argv[2]) if len(sys.argv) > 2 else 97
H = int(sys.argv[3]) if len(sys.argv) > 3 else 61
frames = int(sys.argv[4]) if len(sys.argv) > 4 else 16
scene = prt.HostScene(variant + ".json")
cfg = scene.config()
cam = prt.default_camera(W, H)
seeds = prt.seed_pairs(frames)
out = []
for pool in (0, 1):
    r = prt.Renderer(cfg, device=0)
    r.upload_scene(scene); r.set_camera(cam); r.resize(W, H)
    r.set_option("pool", pool)
    r.render_frames(seeds)
    st = r.read_state().view(prt.PATH_STATE_DTYPE).reshape(H, W)
    print("pool", pool, r.kernel_variant(), "kernel ms", r.stats().kernel_ms)
    out.append(st)
    r.close()
a, b = out
for f in ("samples", "total", "acc", "origin", "mask"):
    d = (a[f].view(np.uint32) != b[f].view(np.uint32))
    if d.ndim == 3: d = d.any(axis=2)
    print(f, "differs in", int(d.sum()), "of", W * H, "pixels")
d = (a["acc"].view(np.uint32) != b["acc"].view(np.uint32)).any(axis=2)
ys, xs = np.nonzero(d)
print("first differing pixels:", list(zip(xs[:12].tolist(), ys[:12].tolist())))
print("segments: ref", float(a["acc"][..., 3].sum()), "pool", float(b["acc"][..., 3].sum()))
print("samples: ref", int(a["samples"].sum()), "pool", int(b["samples"].sum()))
for (x, y) in list(zip(xs[:4].tolist(), ys[:4].tolist())):
    print((x, y), "ref acc", a["acc"][y, x], "samples", a["samples"][y, x], "| pool acc", b["acc"][y, x], "samples", b["samples"][y, x])
# rows of the tile map: fraction of differing pixels per 8x8 tile
th, tw = (H + 7) // 8, (W + 7) // 8
for ty in range(th):
    print("".join("%1d" % min(9, int(10 * d[ty*8:ty*8+8, tx*8:tx*8+8].mean())) for tx in range(tw)))
