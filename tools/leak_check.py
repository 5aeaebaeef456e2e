#!/usr/bin/env python3
"""development helper: create / render / destroy many contexts; device memory must come back"""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
prt = importlib.import_module("photorealistic-rendering-using-opencl_amd")
scene = prt.HostScene("cornell_coat.json"); cfg = scene.config(); seeds = prt.seed_pairs(64)
free0 = torch.cuda.mem_get_info()[0]
for i in range(300):
    r = prt.Renderer(cfg, device=0); r.upload_scene(scene); r.set_camera(prt.default_camera(160, 120)); r.resize(160, 120)
    if i % 2: r.render_spp(2, seeds)
    else: r.render_frames(seeds[:32])
    r.read_framebuffer(); r.close()
    if i % 100 == 99: print(i + 1, "contexts, free delta MB", (free0 - torch.cuda.mem_get_info()[0]) / 1e6, flush=True)
