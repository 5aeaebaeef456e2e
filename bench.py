#!/usr/bin/env python3
"""bench.py -- headline benchmark of the hot path (BASELINE.json):
Msamples/s (width x height x spp / s) at 1920x1080 on scenes/cornell (Lambert + area light),
1024 spp, plus the achieved algorithmic HBM GB/s of the render kernel against the gfx950 peak.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--spp S] [--width X --height Y]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A *step* is one complete render of the workload: reset, then frame-synchronous segments until
every pixel has finished its S-th path (the reference's "N spp" in its own work accounting,
SURVEY.md s8d).  Inputs (scene, BVH, camera, seed table) are resident in HBM before the timed
region.  With N > 1 the frame is split into interleaved row blocks, one set per rank (pixels are
independent; seeds use global pixel coordinates, so the union is bit-identical to the 1-GPU
image of that frame), and ONE RCCL reduce(sum) of the zero-padded full-size framebuffer per step
merges them on rank 0 -- inside the timed region.

Scaling.  Default "weak": per-GPU work is fixed -- the N-GPU frame has N x the pixels of the base
frame (same scene, camera and aspect, both dimensions x sqrt(N): 1920x1080, 2720x1530, 3840x2160,
5424x3051), so every rank renders ~2.07 M pixels x spp as the 1-GPU run does.  `--scaling strong`
keeps the base frame and splits it N ways instead; at 1080p that stops scaling early for a
structural reason measured in DESIGN.md s5: the frame has 32 400 waves of pixels, eight MI355X hold
32 768 resident waves, so the run time falls to the sequential chain of the slowest tile
(spp x path length segments, one after the other), not to work / N.

The JSON line also carries
  roofline      the dominant kernel (render_kernel) priced in ALGORITHMIC bytes: 240 B per pixel
                segment (112 B state read + 112 B state write + 16 B image write, what the
                reference moves per work-item per launch) x segments executed, divided by the
                kernel time measured live with HIP events (libprt does it: `kernel_wall_ms` around
                all the launches of a step on the caller's stream, `avg_launch_ms` around each
                launch on the internal stream it ran on -- two launches, covering interleaved sets
                of tiles, are in flight at a time).  peak = 8 TB/s HBM3E.  `traffic` = measured HBM
                bytes per launch from the rocprofv3 PMC passes under profiles/ (null if absent).
  cpu_baseline  oracle/pt_oracle.c (a scalar-per-pixel CPU port, multi-threaded over pixels) on a
                bounded sample of the same workload, rank 0, N = 1 only.
"""
import argparse
import importlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
PKG_NAME = "photorealistic-rendering-using-opencl_amd"

ALGO_BYTES_PER_SEGMENT = 240
HBM_PEAK_GBPS = 8000.0                # MI355X_MICROARCH.md: 8.0 TB/s spec


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--spp", type=int, default=1024)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--scene", default="cornell_diffuse.json")
    ap.add_argument("--env", default="", choices=["", "sky"], help="sky = the procedural 1024x512 HDR stand-in (configs 3, 4)")
    ap.add_argument("--phase", default="isotropic", choices=["isotropic", "hg", "rayleigh"], help="phase function of the global medium (config 4)")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="N > 1: weak = N x the pixels (frame x sqrt(N) per dimension), strong = the base frame split N ways")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--verify", action="store_true",
                    help="N > 1: after the timed steps rank 0 renders the whole frame alone and compares it bit for bit with the merged one")
    a = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        if world == 1 and a.gpus > 1:
            raise SystemExit("bench.py --gpus %d must be launched with torch.distributed.run --nproc-per-node %d" % (a.gpus, a.gpus))
        a.gpus = world
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: libprt has no CPU fallback")
    # rehearsal knobs for a 1-GPU box: PRT_BENCH_BACKEND=gloo + PRT_BENCH_ONE_DEVICE=1 run N ranks on cuda:0
    one_device = os.environ.get("PRT_BENCH_ONE_DEVICE") == "1"
    backend = os.environ.get("PRT_BENCH_BACKEND", "nccl")
    if one_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend=backend)

    prt = importlib.import_module(PKG_NAME)
    par = importlib.import_module(PKG_NAME + ".parallel")
    import __graft_entry__ as ge
    if not os.path.exists(os.path.join(ROOT, PKG_NAME, "libprt.so")):
        ge.build()

    W, H, spp = a.width, a.height, a.spp
    if world > 1 and a.scaling == "weak":
        W = int(round(a.width * world ** 0.5 / 16.0)) * 16
        H = int(round(a.height * W / float(a.width)))
    if "dragon" in a.scene:
        prt.ensure_dragon_standin()
    scene = prt.HostScene(a.scene)
    cfg = scene.config()
    cfg.phase_function = {"isotropic": 0, "hg": 1, "rayleigh": 2}[a.phase]
    cam = prt.default_camera(W, H)
    max_frames = max(64, spp * max(cfg.max_bounces, 8) + 64)      # a path has at most max_bounces (+1) segments
    seeds = prt.seed_pairs(max_frames)

    r = prt.Renderer(cfg, device=local_rank)
    r.upload_scene(scene)
    if a.env == "sky":
        r.upload_envmap(prt.make_sky(1024, 512))
    r.set_camera(cam)
    # all GPU work of a step -- libprt's launches, the device-to-device copy of the rows, torch's merge -- is ordered
    # on ONE explicit stream (torch's default stream is the null handle, which libprt takes as "use your own")
    stream = torch.cuda.Stream()
    torch.cuda.set_stream(stream)
    r.set_stream(stream.cuda_stream)
    if world == 1:
        r.resize(W, H)
        my_rows = None
    else:
        my_rows = par.rows_of_rank(H, world, rank)
        r.set_row_blocks(W, H, par.BLOCK_ROWS, world, rank)
        tile = torch.zeros((len(my_rows), W, 4), dtype=torch.float32, device="cuda")

    merged = [None]
    kernel_ms = 0.0          # wall time of the GPU work of the timed steps (HIP events around all of it)
    kernel_sum_ms = 0.0      # sum of the durations of the individual launches (HIP events around each)
    launches = 0
    concurrent = 1

    def step(timed):
        nonlocal kernel_ms, kernel_sum_ms, launches, concurrent
        r.reset()
        r.render_spp(spp, seeds)
        if world > 1:
            r.copy_framebuffer_to_device(tile.data_ptr())
            merged[0] = par.merge_on_rank0(tile, my_rows, H, W, dist)
        if timed:
            st = r.stats()
            kernel_ms += st.kernel_ms
            kernel_sum_ms += st.kernel_sum_ms
            launches += st.launches
            concurrent = st.concurrent

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(a.warmup):
        step(False)
    sync()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step(True)
    sync()
    dt = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())

    verified = None
    if a.verify and world > 1:
        torch.cuda.synchronize()
        if rank == 0:
            r1 = prt.Renderer(cfg, device=local_rank)
            r1.upload_scene(scene)
            if a.env == "sky":
                r1.upload_envmap(prt.make_sky(1024, 512))
            r1.set_camera(cam)
            r1.resize(W, H)
            r1.render_spp(spp, seeds)
            alone = r1.read_framebuffer()
            r1.close()
            verified = bool(np.array_equal(alone.view(np.uint32), merged[0].cpu().numpy().view(np.uint32)))
        dist.barrier()

    counts = r.counts(spp)
    seg = torch.tensor([float(counts.segments), float(counts.samples), kernel_ms, float(launches)], dtype=torch.float64, device="cuda")
    if world > 1:
        seg_sum = seg.clone()
        dist.all_reduce(seg_sum, op=dist.ReduceOp.SUM)
        seg_max = seg.clone()
        dist.all_reduce(seg_max, op=dist.ReduceOp.MAX)
        total_segments, total_samples = float(seg_sum[0]), float(seg_sum[1])
        kernel_ms_max = float(seg_max[2])
    else:
        total_segments, total_samples = float(seg[0]), float(seg[1])
        kernel_ms_max = kernel_ms

    if rank == 0:
        steps = max(a.steps, 1)
        msamples = W * H * spp * steps / dt / 1e6
        # rank-0 kernel: its own segments per step over its own kernel time
        own_segments = float(counts.segments)
        achieved = ALGO_BYTES_PER_SEGMENT * own_segments * steps / (kernel_ms * 1e-3) / 1e9 if kernel_ms > 0 else 0.0
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "r01_traffic.json")
        if os.path.exists(tpath):
            try:
                tj = json.load(open(tpath))
                if tj.get("workload") == "%dx%d_%dspp_%s" % (a.width, a.height, spp, a.scene):   # per launch of the base frame = per rank
                    traffic = tj.get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        out = {
            "metric": "Msamples/s (width x height x spp / s) at %dx%d" % (a.width, a.height),
            "value": round(msamples, 3),
            "unit": "Msamples/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(dt / steps * 1e3, 3),
            "higher_is_better": True,
            "scaling": a.scaling if world > 1 else "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": "%s %dx%d %dspp%s%s, %d x MI355X" % (
                           "scenes/cornell (teapot DIFF = Lambert + sphere area light)" if a.scene == "cornell_diffuse.json" else a.scene,
                           W, H, spp, " + procedural HDR env" if a.env else "", " HG phase" if a.phase == "hg" else "", world),
                       "scene": a.scene, "width": W, "height": H, "spp": spp,
                       "mean_path_length": round(total_segments / max(total_samples, 1.0), 4),
                       "segments_per_step": total_segments,
                       "base_frame": "%dx%d" % (a.width, a.height),
                       "parallelism": "single GPU" if world == 1 else "%s scaling: %dx%d frame in interleaved 16-row blocks over %d ranks + 1 RCCL reduce" % (a.scaling, W, H, world)},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBPS, 5), "traffic": traffic,
                         "kernel": "render_kernel", "algorithmic_bytes_per_segment": ALGO_BYTES_PER_SEGMENT,
                         "launches": launches, "avg_launch_ms": round(kernel_sum_ms / max(launches, 1), 4),
                         "concurrent_launches": concurrent, "kernel_wall_ms": round(kernel_ms, 3),
                         "note": "libprt keeps %d launches of render_kernel in flight (interleaved sets of tiles on internal streams): "
                                 "`achieved` = algorithmic bytes of all launches / wall time of the GPU work (HIP events around it); "
                                 "`avg_launch_ms` = mean duration of one launch (HIP events around each, = the profiler's average), "
                                 "launches x avg_launch_ms ~ %d x kernel_wall_ms" % (concurrent, concurrent),
                         "gsegments_per_s": round(own_segments * steps / (kernel_ms * 1e-3) / 1e9, 4) if kernel_ms > 0 else 0.0},
        }
        if verified is not None:
            out["config"]["merged_frame_equals_single_gpu_render"] = verified
        if world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(prt, a)
            ref = cpu_reference_build(prt, a)
            if ref is not None:
                out["cpu_reference_build"] = ref
        print(json.dumps(out))
    r.close()
    if world > 1:
        dist.destroy_process_group()


def cpu_baseline(prt, a):
    """oracle/pt_oracle.c timed on the host cores: 960x540 (1/2 of the frame in each dimension,
    same camera) at 192 spp of the same scene -- about 15 s of CPU work on the GPU box."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle_api as O
    W, H, spp = 960, 540, 192
    try:
        cores = len(os.sched_getaffinity(0))
    except Exception:
        cores = os.cpu_count() or 1
    cores = max(1, min(cores, 64))
    scene = prt.HostScene(a.scene)
    cfg = scene.config()
    cfg.phase_function = {"isotropic": 0, "hg": 1, "rayleigh": 2}[a.phase]
    cam = prt.default_camera(W, H)
    seeds = prt.seed_pairs(spp * max(cfg.max_bounces, 8) + 64)
    env = prt.make_sky(1024, 512) if a.env == "sky" else None
    rs = O.Restatement()
    t0 = time.perf_counter()
    state, _ = rs.render(cfg, scene.desc, cam, W, H, seeds, env=env, spp_limit=spp, threads=cores)
    dt = time.perf_counter() - t0
    assert (state["samples"] == spp).all()
    return {"value": round(W * H * spp / dt / 1e6, 4), "unit": "Msamples/s", "cores": cores, "kind": "port",
            "sample": "%dx%d %dspp of the same scene and camera (oracle/pt_oracle.c, %d threads, %.1f s)" % (W, H, spp, cores, dt),
            "gsegments_per_s": round(float(state["acc"][:, 3].sum()) / dt / 1e9, 5)}


def cpu_reference_build(prt, a):
    """the reference's OWN kernel text compiled for the host (oracle/_ref, built in the development container
    and shipped as a binary) on a smaller sample of the same workload; None if that build is not present"""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle_api as O
    variant = os.path.splitext(a.scene)[0]
    if a.phase != "isotropic":
        variant += "_" + a.phase
    if not O.ref_available(variant):
        return None
    W, H, spp = 480, 270, 48
    try:
        cores = len(os.sched_getaffinity(0))
    except Exception:
        cores = os.cpu_count() or 1
    cores = max(1, min(cores, 64))
    scene = prt.HostScene(a.scene)
    cfg = scene.config()
    cam = prt.default_camera(W, H)
    seeds = prt.seed_pairs(spp * max(cfg.max_bounces, 8) + 64)
    env = prt.make_sky(1024, 512) if a.env == "sky" else None
    t0 = time.perf_counter()
    state, _ = O.RefOracle(variant).render(scene.desc, bytes(cam), W, H, seeds, env=env, spp_limit=spp, threads=cores)
    dt = time.perf_counter() - t0
    if not (state["samples"] == spp).all():
        return None
    return {"value": round(W * H * spp / dt / 1e6, 4), "unit": "Msamples/s", "cores": cores, "kind": "reference",
            "sample": "%dx%d %dspp of the same scene and camera (reference kernel text, clang x86-64 + prt_detmath, %d threads, %.1f s)" % (W, H, spp, cores, dt)}


if __name__ == "__main__":
    main()
