#!/usr/bin/env python3
"""development helper: the PCIe-inclusive rate of the headline config -- scene upload + render + framebuffer read-back"""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
prt = importlib.import_module("photorealistic-rendering-using-opencl_amd")
W, H, spp = 1920, 1080, int(sys.argv[1]) if len(sys.argv) > 1 else 1024
scene = prt.HostScene("cornell_diffuse.json"); cfg = scene.config(); seeds = prt.seed_pairs(spp * 12 + 64)
r = prt.Renderer(cfg, device=0)
for it in range(2):
    t0 = time.time(); r.upload_scene(scene); r.set_camera(prt.default_camera(W, H)); r.resize(W, H); r.synchronize(); t1 = time.time()
    r.render_spp(spp, seeds); r.synchronize(); t2 = time.time()
    img = r.read_framebuffer(); t3 = time.time()
print("upload+resize %.1f ms, render %.1f ms, read_framebuffer (33 MB) %.1f ms -> %.1f Msamples/s resident, %.1f PCIe-inclusive"
      % ((t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3, W * H * spp / (t2 - t1) / 1e6, W * H * spp / (t3 - t0) / 1e6))
