"""development helper: G segments/s of prt_render_frames (every pixel does every frame: no samples-per-pixel tail), to price the tail of prt_render_spp"""
import sys, time, importlib
sys.path.insert(0, '/root/repo')
import numpy as np, torch
prt = importlib.import_module("photorealistic-rendering-using-opencl_amd")
W, H = 1920, 1080
scene = prt.HostScene("cornell_diffuse.json"); cfg = scene.config(); cam = prt.default_camera(W, H)
frames = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
seeds = prt.seed_pairs(frames)
r = prt.Renderer(cfg, device=0); r.upload_scene(scene); r.set_camera(cam); r.resize(W, H)
for rep in range(2):
    r.reset()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    r.render_frames(seeds)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print("frame mode: %d frames, %.3f s, %.3f G segments/s" % (frames, dt, W * H * frames / dt / 1e9))
