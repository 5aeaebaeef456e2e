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
image of that frame), and ONE RCCL collective per step -- a gather of every rank's rows to rank 0 --
puts the framebuffer together there, inside the timed region.  By default rank 0 then renders
the whole frame alone (untimed) and the JSON line says whether the merged frame equals it bit for bit.

Scaling.  The N-GPU line is BASELINE's own frame: the FIXED 1920x1080 x 1024 spp frame of config 2 split N ways (strong scaling;
`metric` names the frame that was rendered, `scaling` says "strong"), every rank its interleaved row blocks, one gather per step,
the merged frame checked against the one GPU's.  That stops scaling early for a structural reason measured in DESIGN.md s5: the
frame is 32 400 waves of pixels, one MI355X holds 6 144, so from N = 4 on a rank's share is one round of waves and its run time the
sequential chain of its slowest pixel (spp x path length segments, one after the other), not work / N.  Pixels are independent
units, so the path shards with no data-path exchange and its WEAK scaling -- fixed work per GPU: the same view with N x the pixels
(both dimensions x sqrt(N): 2720x1530, 3840x2160, 5424x3051; same camera and spp, hence the same cost per pixel) -- rides along as
the sub-record `weak` (one step).  `--scaling weak` makes that the headline line instead (its `metric` then names the bigger
frame).  With N = 8 the run adds BASELINE config 5 -- the 871 k-triangle stand-in at 3840x2160, 8-way split -- as `config5`, verified.

The JSON line also carries
  roofline      the dominant kernel (render_kernel) priced in ALGORITHMIC bytes: 240 B per pixel
                segment (112 B state read + 112 B state write + 16 B image write, what the
                reference moves per work-item per launch) x segments executed, divided by the
                kernel time measured live with HIP events (libprt does it: `kernel_wall_ms` around
                all the launches of a step on the caller's stream, `avg_launch_ms` around each
                launch on the internal stream it ran on -- two launches, covering interleaved sets
                of tiles, are in flight at a time).  peak = 8 TB/s HBM3E.  `traffic` = measured HBM
                bytes per launch from the rocprofv3 PMC passes under profiles/ (null if absent), and
                `measured_hbm_gbps` what that is per second of this run.  The path state lives in
                registers for the 512 frames of a launch, so the measured traffic is ~2 % of the
                algorithmic bytes: HBM is the bound SURVEY s8d prescribes for the accounting, not what
                binds.  What does is in `valu`: vector instructions per wave-segment and lane occupancy
                (PMC passes of the same workload, profiles/r04_pmc_render_kernel.json) and, from this
                run's segment rate, the share of the chip's vector issue slots they take.
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


def run_workload(prt, par, torch, dist, np, a, world, rank, local_rank, scene_name, base_w, base_h, spp, env_name, phase, scaling, steps, warmup, verify):
    """times `steps` renders of one workload on this rank's share of the frame; returns the measurements (rank 0: all of them)"""
    W, H = base_w, base_h
    if world > 1 and scaling == "weak":
        W = int(round(base_w * world ** 0.5 / 16.0)) * 16
        H = int(round(base_h * W / float(base_w)))
    if "dragon" in scene_name:
        if rank == 0:
            prt.ensure_dragon_standin()                    # 62 MB, generated once; the other ranks wait for the file
        if world > 1:
            dist.barrier()
    scene = prt.HostScene(scene_name)
    cfg = scene.config()
    cfg.phase_function = {"isotropic": 0, "hg": 1, "rayleigh": 2}[phase]
    cam = prt.default_camera(W, H)
    max_frames = max(64, spp * max(cfg.max_bounces, 8) + 64)      # a path has at most max_bounces (+1) segments
    seeds = prt.seed_pairs(max_frames)

    r = prt.Renderer(cfg, device=local_rank)
    r.upload_scene(scene)
    if env_name == "sky":
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
        tile = torch.zeros((par.max_rows_per_rank(H, world), W, 4), dtype=torch.float32, device="cuda")

    merged = [None]
    m = {"kernel_ms": 0.0, "kernel_sum_ms": 0.0, "launches": 0, "concurrent": 1}

    def step(timed):
        r.reset()
        r.render_spp(spp, seeds)
        if world > 1:
            r.copy_framebuffer_to_device(tile.data_ptr())          # this rank's rows (the padding rows stay zero)
            merged[0] = par.merge_on_rank0(tile, H, W, world, dist)
        if timed:
            st = r.stats()
            m["kernel_ms"] += st.kernel_ms
            m["kernel_sum_ms"] += st.kernel_sum_ms
            m["launches"] += st.launches
            m["concurrent"] = st.concurrent

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(warmup):
        step(False)
    sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        step(True)
    sync()
    dt = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())

    verified = None
    if verify and world > 1:
        torch.cuda.synchronize()
        if rank == 0:
            r1 = prt.Renderer(cfg, device=local_rank)
            r1.upload_scene(scene)
            if env_name == "sky":
                r1.upload_envmap(prt.make_sky(1024, 512))
            r1.set_camera(cam)
            r1.resize(W, H)
            r1.render_spp(spp, seeds)
            alone = r1.read_framebuffer()
            r1.close()
            verified = bool(np.array_equal(alone.view(np.uint32), merged[0].cpu().numpy().view(np.uint32)))
        dist.barrier()

    variant = r.kernel_variant()
    counts = r.counts(spp)
    seg = torch.tensor([float(counts.segments), float(counts.samples)], dtype=torch.float64, device="cuda")
    if world > 1:
        dist.all_reduce(seg, op=dist.ReduceOp.SUM)
    r.close()
    torch.cuda.set_stream(torch.cuda.default_stream())
    steps_ = max(steps, 1)
    own_segments = float(counts.segments)
    return {"W": W, "H": H, "spp": spp, "dt": dt, "msamples": W * H * spp * steps_ / dt / 1e6, "ms_per_step": dt / steps_ * 1e3,
            "total_segments": float(seg[0]), "total_samples": float(seg[1]), "own_segments": own_segments, "verified": verified,
            "kernel_ms": m["kernel_ms"], "kernel_sum_ms": m["kernel_sum_ms"], "launches": m["launches"], "concurrent": m["concurrent"],
            "max_bounces": cfg.max_bounces, "variant": variant}


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
    ap.add_argument("--scaling", default="strong", choices=["strong", "weak"],
                    help="N > 1: strong (default) = the FIXED base frame split N ways (BASELINE's frame); weak = N x the pixels (frame x sqrt(N) per dimension, fixed work per GPU)")
    ap.add_argument("--no-weak", action="store_true",
                    help="N > 1: skip the extra one-step run of the weak-scaling frame (`weak`)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-verify", action="store_true",
                    help="N > 1: skip the check that rank 0's merged frame equals, bit for bit, the frame one GPU renders alone")
    ap.add_argument("--config5-spp", type=int, default=256,
                    help="N = 8 only: spp of the extra BASELINE config 5 run (871 k-triangle stand-in, 3840x2160, 8-way split); 0 = skip")
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
    # what is timed is the library the working tree builds, or nothing: a libprt.so left behind by other sources (round 3: an experiment's
    # kernel, twice) is refused before anything runs -- needs no GPU.  (A missing library is built below, from the tree.)
    if os.path.exists(os.path.join(ROOT, PKG_NAME, "libprt.so")) or os.environ.get("PRT_LIB"):
        pkg0 = importlib.import_module(PKG_NAME)
        try:
            pkg0.check_build_id()
        except pkg0.StaleLibrary as e:
            sys.stderr.write("bench.py: refusing to time a stale library: %s\n" % e)
            raise SystemExit(3)
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

    # one rank builds (the others would write the same object files at the same time), everybody waits
    if not os.path.exists(os.path.join(ROOT, PKG_NAME, "libprt.so")):
        if rank == 0:
            import __graft_entry__ as ge
            ge.build()
    if world > 1:
        dist.barrier()
    prt = importlib.import_module(PKG_NAME)
    par = importlib.import_module(PKG_NAME + ".parallel")

    verify = world > 1 and not a.no_verify
    res = run_workload(prt, par, torch, dist, np, a, world, rank, local_rank, a.scene, a.width, a.height, a.spp, a.env, a.phase,
                       a.scaling, a.steps, a.warmup, verify)
    # beside the line of BASELINE's frame: the other kind of scaling, one step
    res_other = None
    if world > 1 and not a.no_weak:
        res_other = run_workload(prt, par, torch, dist, np, a, world, rank, local_rank, a.scene, a.width, a.height, a.spp, a.env, a.phase,
                                 "weak" if a.scaling == "strong" else "strong", 1, 1, False)
    # BASELINE config 5 is defined on 8 GPUs: the 871 k-triangle stand-in at 3840x2160, 8-way tile split + framebuffer merge
    res5 = None
    if (world == 8 or (world > 1 and os.environ.get("PRT_BENCH_CONFIG5") == "1")) and a.config5_spp > 0 and a.scene == "cornell_diffuse.json":
        res5 = run_workload(prt, par, torch, dist, np, a, world, rank, local_rank, "cornell_dragon.json", 3840, 2160, a.config5_spp, "", "isotropic",
                            "strong", 1, 1, verify)

    if rank == 0:
        W, H, spp = res["W"], res["H"], res["spp"]
        steps = max(a.steps, 1)
        kernel_ms, kernel_sum_ms, launches, concurrent = res["kernel_ms"], res["kernel_sum_ms"], res["launches"], res["concurrent"]
        total_segments, total_samples, own_segments = res["total_segments"], res["total_samples"], res["own_segments"]
        # rank-0 kernel: its own segments per step over its own kernel time
        achieved = ALGO_BYTES_PER_SEGMENT * own_segments * steps / (kernel_ms * 1e-3) / 1e9 if kernel_ms > 0 else 0.0
        traffic = None
        for tname in ("r04_traffic.json", "r03_traffic.json", "r02_traffic.json", "r01_traffic.json"):
            tpath = os.path.join(ROOT, "profiles", tname)
            if traffic is None and os.path.exists(tpath):
                try:
                    tj = json.load(open(tpath))
                    if world == 1 and tj.get("workload") == "%dx%d_%dspp_%s" % (a.width, a.height, spp, a.scene):   # per launch of the whole frame
                        traffic = tj.get("hbm_bytes_per_launch")
                except Exception:
                    traffic = None
        avg_launch_ms = kernel_sum_ms / max(launches, 1)
        gseg = own_segments * steps / (kernel_ms * 1e-3) / 1e9 if kernel_ms > 0 else 0.0
        valu = None
        ppath = os.path.join(ROOT, "profiles", "r04_pmc_render_kernel.json")
        if not os.path.exists(ppath):
            ppath = os.path.join(ROOT, "profiles", "r03_pmc_render_kernel.json")
        if world == 1 and os.path.exists(ppath):
            try:
                pj = json.load(open(ppath))
                if pj.get("workload_key") == "%dx%d_%s" % (a.width, a.height, a.scene):
                    d = pj["derived"]
                    vws = d["valu_per_wave_segment"]
                    # 256 CUs x 4 SIMDs, one wave64 vector instruction per 2 cycles per SIMD at 2.4 GHz (MI355X_MICROARCH.md: wave scheduling)
                    issue_peak = 1024 * 2.4e9 / 2.0
                    valu = {"valu_per_wave_segment": round(vws, 1), "salu_per_wave_segment": round(d["salu_per_wave_segment"], 1),
                            "lane_occupancy": round(d["lane_occupancy = SQ_THREAD_CYCLES_VALU/(64*SQ_ACTIVE_INST_VALU)"], 4),
                            "wait_fraction": round(d["wait_fraction = SQ_WAIT_ANY/SQ_WAVE_CYCLES"], 4),
                            "issue_frac": round(vws * (gseg * 1e9 / 64.0) / issue_peak, 4),
                            "source": "profiles/" + os.path.basename(ppath) + " (rocprofv3 --pmc passes of this workload at %s spp); issue_frac = "
                                      "valu_per_wave_segment x this run's wave-segments/s / (1024 SIMDs x 2.4 GHz / 2)" % pj.get("spp", "?")}
            except Exception:
                valu = None
        out = {
            "metric": "Msamples/s (width x height x spp / s) at %dx%d" % (W, H),
            "value": round(res["msamples"], 3),
            "unit": "Msamples/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(res["ms_per_step"], 3),
            "higher_is_better": True,
            "scaling": a.scaling,
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "build_id": prt.build_id(),
            "config": {"workload": "%s %dx%d %dspp%s%s, %d x MI355X" % (
                           "scenes/cornell (teapot DIFF = Lambert + sphere area light)" if a.scene == "cornell_diffuse.json" else a.scene,
                           W, H, spp, " + procedural HDR env" if a.env else "", " HG phase" if a.phase == "hg" else "", world),
                       "scene": a.scene, "width": W, "height": H, "spp": spp,
                       "mean_path_length": round(total_segments / max(total_samples, 1.0), 4),
                       "segments_per_step": total_segments,
                       "base_frame": "%dx%d" % (a.width, a.height),
                       "parallelism": "single GPU" if world == 1 else "%s scaling: %dx%d frame in interleaved 16-row blocks over %d ranks + 1 RCCL gather of the rows to rank 0 per step" % (a.scaling, W, H, world)},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBPS, 5), "traffic": traffic,
                         "measured_hbm_gbps": (round(traffic / (avg_launch_ms * 1e-3) * concurrent / 1e9, 2) if traffic and avg_launch_ms > 0 else None),
                         "binds": "vector issue at about a third of the lanes + memory latency of the BVH walk, not HBM (see valu)",
                         "valu": valu,
                         "kernel": res["variant"], "algorithmic_bytes_per_segment": ALGO_BYTES_PER_SEGMENT,
                         "launches": launches, "avg_launch_ms": round(avg_launch_ms, 4),
                         "concurrent_launches": concurrent, "kernel_wall_ms": round(kernel_ms, 3),
                         "note": "libprt keeps %d launches of render_kernel in flight (interleaved sets of tiles on internal streams): "
                                 "`achieved` = algorithmic bytes of all launches / wall time of the GPU work (HIP events around it); "
                                 "`avg_launch_ms` = mean duration of one launch (HIP events around each, = the profiler's average), "
                                 "launches x avg_launch_ms ~ %d x kernel_wall_ms" % (concurrent, concurrent),
                         "gsegments_per_s": round(gseg, 4)},
        }
        if res["verified"] is not None:
            out["config"]["merged_frame_equals_single_gpu_render"] = res["verified"]
        if res_other is not None:
            key = "weak" if a.scaling == "strong" else "fixed_frame"
            out[key] = {"workload": "%s %dx%d %dspp over %d ranks (%s, one step)" % (
                            a.scene, res_other["W"], res_other["H"], a.spp, world,
                            "weak scaling: %d x the pixels of the base frame, fixed work per GPU" % world if key == "weak" else "strong scaling of the base frame"),
                        "value": round(res_other["msamples"], 3), "unit": "Msamples/s", "ms_per_step": round(res_other["ms_per_step"], 3)}
        if res5 is not None:
            out["config5"] = {"workload": "scenes/cornell_dragon (871 k-triangle stand-in) 3840x2160 %dspp, %d x MI355X tile split + RCCL framebuffer merge "
                                          "(BASELINE config 5 at a reduced spp: the full 8192 spp is %d x this work)" % (res5["spp"], world, 8192 // max(res5["spp"], 1)),
                              "value": round(res5["msamples"], 3), "unit": "Msamples/s", "ms_per_step": round(res5["ms_per_step"], 3),
                              "mean_path_length": round(res5["total_segments"] / max(res5["total_samples"], 1.0), 4),
                              "gsegments_per_s_all_gpus": round(res5["total_segments"] / (res5["dt"]) / 1e9, 4),
                              "merged_frame_equals_single_gpu_render": res5["verified"]}
        if world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(prt, a)
            ref = cpu_reference_build(prt, a)
            if ref is not None:
                out["cpu_reference_build"] = ref
        print(json.dumps(out))
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
