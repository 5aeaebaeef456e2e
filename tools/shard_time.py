#!/usr/bin/env python3
"""development helper: times the row-block shard one rank of an N-GPU run would render, on ONE GPU, to predict the
strong-scaling efficiency of bench.py (no collective; the reduce of 33 MB is negligible next to the render)
usage: shard_time.py [spp=1024] [scene=cornell_diffuse.json] [width=1920] [height=1080]
       (BASELINE config 5: shard_time.py 64 cornell_dragon.json 3840 2160)"""
import importlib
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
prt = importlib.import_module("photorealistic-rendering-using-opencl_amd")
par = importlib.import_module("photorealistic-rendering-using-opencl_amd.parallel")
spp = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
scene_name = sys.argv[2] if len(sys.argv) > 2 else "cornell_diffuse.json"
W, H = (int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else (1920, 1080)
if "dragon" in scene_name:
    prt.ensure_dragon_standin()
scene = prt.HostScene(scene_name)
cfg = scene.config()
seeds = prt.seed_pairs(spp * max(cfg.max_bounces, 8) + 64)
base = None
for world in (1, 2, 4, 8):
    worst = 0.0
    for rank in sorted({0, world // 2, world - 1}):
        r = prt.Renderer(cfg, device=0)
        r.upload_scene(scene)
        r.set_camera(prt.default_camera(W, H))
        if world == 1:
            r.resize(W, H)
        else:
            r.set_row_blocks(W, H, par.BLOCK_ROWS, world, rank)
        for it in range(2):
            r.reset()
            r.synchronize()
            t0 = time.time()
            used = r.render_spp(spp, seeds)
            r.synchronize()
            dt = time.time() - t0
        worst = max(worst, dt)
        st = r.stats()
        print("   world %d rank %d: %.3f s wall, kernel %.1f ms over %d launches (whole context)" % (world, rank, dt, st.kernel_ms, st.launches), flush=True)
        r.close()
    base = base or worst
    print("world %d: slowest sampled rank %.3f s, frames %d, efficiency vs 1 GPU %.3f" % (world, worst, used, base / (world * worst)), flush=True)
