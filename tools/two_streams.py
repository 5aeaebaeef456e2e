#!/usr/bin/env python3
"""development helper: does rendering the frame as K independent row-block parts on K streams of ONE GPU hide the
drain at the end of every launch?  (K contexts, one thread each)"""
import importlib
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
prt = importlib.import_module("photorealistic-rendering-using-opencl_amd")
par = importlib.import_module("photorealistic-rendering-using-opencl_amd.parallel")
spp = int(sys.argv[1]) if len(sys.argv) > 1 else 256
W, H = 1920, 1080
scene = prt.HostScene("cornell_diffuse.json")
cfg = scene.config()
seeds = prt.seed_pairs(spp * 8 + 64)
for K in (1, 2, 3, 4):
    rs = []
    for k in range(K):
        r = prt.Renderer(cfg, device=0)
        r.upload_scene(scene)
        r.set_camera(prt.default_camera(W, H))
        if K == 1:
            r.resize(W, H)
        else:
            r.set_row_blocks(W, H, par.BLOCK_ROWS, K, k)
        rs.append(r)
    def work(r):
        r.reset()
        r.render_spp(spp, seeds)
        r.synchronize()
    for it in range(2):
        ts = [threading.Thread(target=work, args=(r,)) for r in rs]
        t0 = time.time()
        for t in ts: t.start()
        for t in ts: t.join()
        dt = time.time() - t0
    print("K=%d contexts/streams: %.3f s  (%.1f Msamples/s)" % (K, dt, W * H * spp / dt / 1e6), flush=True)
    for r in rs: r.close()
