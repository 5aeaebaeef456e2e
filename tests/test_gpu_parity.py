"""GPU parity tests: the HIP path behind the C ABI (libprt.so) against the CPU oracle
(oracle/pt_oracle.c) and against the golden fixtures produced by the reference build.
Bar: bit-identical path state (every field of the 112-byte RTD record) and framebuffer."""
import importlib
import os

import numpy as np
import pytest

from conftest import ALPHA_VARIANTS, variant_config, GOLDEN, PKG_NAME, ROOT, VARIANTS, VIEW_VARIANTS, variant_camera

pytestmark = pytest.mark.gpu


def _setup(prt, variant, W, H, rows=None, row0=0):
    scene_json, phase, use_env = VARIANTS[variant]
    scene = prt.HostScene(scene_json)
    cfg = variant_config(scene, variant)
    cfg.phase_function = phase
    cam = variant_camera(prt, variant, W, H)
    env = prt.make_sky(64, 32) if use_env else None
    r = prt.Renderer(cfg, device=0)
    r.upload_scene(scene)
    if env is not None:
        r.upload_envmap(env)
    r.set_camera(cam)
    if rows is None:
        r.resize(W, H)
    else:
        r.set_tile(W, H, row0, rows)
    return scene, cfg, cam, env, r


def _assert_same(oracle, ostate, oimg, state, img, what):
    bad = oracle.state_fields_equal(ostate, state.view(oracle.PATH_STATE_DTYPE))
    assert not bad, "%s: path state differs in %s" % (what, bad)
    assert oracle.images_equal(oimg, img), "%s: framebuffer differs" % what


@pytest.mark.parametrize("variant", list(VARIANTS))
def test_hip_matches_golden(prt, oracle, variant):
    g = np.load(os.path.join(GOLDEN, variant + ".npz"))
    W, H, frames = int(g["width"]), int(g["height"]), int(g["frames"])
    scene, cfg, cam, env, r = _setup(prt, variant, W, H)
    r.render_frames(prt.seed_pairs(frames))
    gstate = np.ascontiguousarray(g["state"]).view(oracle.PATH_STATE_DTYPE).reshape(-1)
    _assert_same(oracle, gstate, g["image"], r.read_state(), r.read_framebuffer(), variant + " vs reference golden")
    r.close()


@pytest.mark.parametrize("variant", list(VARIANTS))
def test_hip_matches_oracle(prt, oracle, variant):
    W, H, frames = 97, 61, 160          # ragged: not a multiple of the 16x16 workgroup tile
    scene, cfg, cam, env, r = _setup(prt, variant, W, H)
    seeds = prt.seed_pairs(frames)
    r.render_frames(seeds)
    ostate, oimg = oracle.Restatement().render(cfg, scene.desc, cam, W, H, seeds, env=env, threads=16)
    _assert_same(oracle, ostate, oimg, r.read_state(), r.read_framebuffer(), variant)
    st = r.counts()
    assert st.samples == int(ostate["samples"].sum()) and st.segments == int(ostate["acc"][:, 3].sum())
    r.close()


def test_spp_mode_matches_golden_and_oracle(prt, oracle):
    g = np.load(os.path.join(GOLDEN, "cornell_diffuse_spp.npz"))
    W, H, maxf, spp = int(g["width"]), int(g["height"]), int(g["frames"]), int(g["spp"])
    scene, cfg, cam, env, r = _setup(prt, "cornell_diffuse", W, H)
    used = r.render_spp(spp, prt.seed_pairs(maxf))
    assert used <= maxf
    state, img = r.read_state(), r.read_framebuffer()
    gstate = np.ascontiguousarray(g["state"]).view(oracle.PATH_STATE_DTYPE).reshape(-1)
    _assert_same(oracle, gstate, g["image"], state, img, "spp golden")
    assert (state["samples"] == spp).all() and (state["reset"] != 0).all()
    st = r.counts(spp)
    assert st.finished_pixels == W * H and st.samples == spp * W * H
    r.close()


def test_frame_batching_is_invisible(prt, oracle):
    """frames 1..n in one call == the same frames in several calls (state carries over exactly)"""
    W, H = 64, 40
    scene, cfg, cam, env, r = _setup(prt, "cornell_coat", W, H)
    seeds = prt.seed_pairs(70)
    r.render_frames(seeds)
    s1, i1 = r.read_state(), r.read_framebuffer()
    r.reset()
    r.render_frames(seeds[:2 * 33], first_frame=1)
    r.render_frames(seeds[2 * 33:2 * 34], first_frame=34)
    r.render_frames(seeds[2 * 34:], first_frame=35)
    _assert_same(oracle, s1.view(oracle.PATH_STATE_DTYPE), i1, r.read_state(), r.read_framebuffer(), "batching")
    r.close()


def test_state_roundtrip_checkpoint(prt, oracle):
    """prt_read_state / prt_write_state: resume from a downloaded checkpoint in a NEW context"""
    W, H = 48, 32
    scene, cfg, cam, env, r = _setup(prt, "cornell_roughcond", W, H)
    seeds = prt.seed_pairs(60)
    r.render_frames(seeds[:2 * 25])
    ck = r.read_state()
    r.render_frames(seeds[2 * 25:], first_frame=26)
    s_full, i_full = r.read_state(), r.read_framebuffer()
    r.close()
    scene, cfg, cam, env, r2 = _setup(prt, "cornell_roughcond", W, H)
    r2.write_state(ck)
    r2.render_frames(seeds[2 * 25:], first_frame=26)
    _assert_same(oracle, s_full.view(oracle.PATH_STATE_DTYPE), i_full, r2.read_state(), r2.read_framebuffer(), "checkpoint")
    r2.close()


def test_row_tiles_union_equals_full_frame(prt, oracle):
    """multi-GPU partitioning: row tiles rendered by separate contexts == the full-frame render"""
    W, H, frames = 80, 50, 64
    scene, cfg, cam, env, r = _setup(prt, "cornell_diffuse", W, H)
    seeds = prt.seed_pairs(frames)
    r.render_frames(seeds)
    full_img, full_state = r.read_framebuffer(), r.read_state()
    r.close()
    parts_i, parts_s = [], []
    for row0, rows in ((0, 13), (13, 20), (33, 17)):
        scene, cfg, cam, env, rt = _setup(prt, "cornell_diffuse", W, H, rows=rows, row0=row0)
        rt.render_frames(seeds)
        parts_i.append(rt.read_framebuffer())
        parts_s.append(rt.read_state())
        rt.close()
    _assert_same(oracle, full_state.view(oracle.PATH_STATE_DTYPE), full_img,
                 np.concatenate(parts_s), np.concatenate(parts_i, axis=0), "row tiles")


def test_interleaved_row_blocks_union_equals_full_frame(prt, oracle):
    """the bench's multi-GPU split: interleaved 16-row blocks per rank"""
    W, H, frames, parts = 64, 77, 48, 3
    scene, cfg, cam, env, r = _setup(prt, "cornell_coat", W, H)
    seeds = prt.seed_pairs(frames)
    r.render_frames(seeds)
    full_img, full_state = r.read_framebuffer(), r.read_state().reshape(H, W)
    r.close()
    got_img = np.zeros_like(full_img)
    got_state = np.zeros_like(full_state)
    for part in range(parts):
        rows = np.array([y for y in range(H) if (y // 16) % parts == part])
        rt = prt.Renderer(cfg, device=0)
        rt.upload_scene(scene)
        rt.set_camera(cam)
        rt.set_row_blocks(W, H, 16, parts, part)
        rt.render_frames(seeds)
        got_img[rows] = rt.read_framebuffer()
        got_state[rows] = rt.read_state().reshape(len(rows), W)
        rt.close()
    _assert_same(oracle, full_state.reshape(-1).view(oracle.PATH_STATE_DTYPE), full_img, got_state.reshape(-1), got_img, "row blocks")


def test_device_side_merge_of_row_block_parts_with_torch(prt, oracle):
    """the multi-GPU merge path of bench.py on one GPU: every part copies its rows device-to-device into a torch
    tensor (prt_copy_framebuffer_to_device) and the rows are scattered into the full frame by row index (what
    parallel.merge_on_rank0 does with the gathered tiles) -- both with the context's own stream and with a torch stream
    handed to prt_set_stream"""
    import torch
    par = importlib.import_module(PKG_NAME + ".parallel")
    W, H, frames, parts = 72, 50, 40, 3
    scene, cfg, cam, env, r = _setup(prt, "cornell_diffuse", W, H)
    seeds = prt.seed_pairs(frames)
    r.render_frames(seeds)
    full = r.read_framebuffer()
    r.close()
    for use_torch_stream in (False, True):
        total = torch.zeros((H, W, 4), dtype=torch.float32, device="cuda")
        stream = torch.cuda.Stream() if use_torch_stream else None
        for part in range(parts):
            rp = prt.Renderer(cfg, device=0)
            rp.upload_scene(scene)
            rp.set_camera(cam)
            if stream is not None:
                rp.set_stream(stream.cuda_stream)
            rp.set_row_blocks(W, H, 16, parts, part)
            rows = par.rows_of_rank(H, parts, part)
            tile = torch.zeros((len(rows), W, 4), dtype=torch.float32, device="cuda")
            if stream is not None:
                stream.wait_stream(torch.cuda.current_stream())          # the zero fill above
            rp.render_frames(seeds)                                      # asynchronous
            rp.copy_framebuffer_to_device(tile.data_ptr())
            if stream is not None:
                torch.cuda.current_stream().wait_stream(stream)
            total.index_copy_(0, torch.as_tensor(np.asarray(rows), dtype=torch.long, device="cuda"), tile)
            torch.cuda.synchronize()
            rp.close()
        assert oracle.images_equal(full, total.cpu().numpy()), "merged parts differ (torch stream: %s)" % use_torch_stream


def test_build_then_smoke_in_one_process_and_torch_after_libprt():
    """PyTorch-ROCm carries its own HIP runtime; two runtimes in one process do not share the device, so the package
    loads torch before libprt.so.  Fresh processes: build() + smoke() back to back, and torch used after a context"""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import __graft_entry__ as g, importlib; g.build(); g.smoke(); "
            "pkg = importlib.import_module(g.PKG_NAME); import torch; assert torch.cuda.is_available(); "
            "r = pkg.Renderer(pkg.HostScene('cornell_coat.json').config(), device=0); r.close(); "
            "assert torch.zeros(4, device='cuda').sum().item() == 0; print('both runtimes fine')")
    out = subprocess.run([sys.executable, "-c", code], cwd=root, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=900)
    assert out.returncode == 0 and "both runtimes fine" in out.stdout, out.stdout[-2000:]


def test_camera_change_and_reset(prt, oracle):
    W, H = 56, 40
    scene, cfg, cam, env, r = _setup(prt, "cornell_coat", W, H)
    seeds = prt.seed_pairs(40)
    r.render_frames(seeds)
    cam2 = prt.orbit_camera(W, H, d_yaw=0.4, d_pitch=0.1, d_radius=-0.2)
    r.set_camera(cam2)
    r.reset()                                   # buffer_reset branch of render(), src/main.cpp:283-291
    r.render_frames(seeds)
    ostate, oimg = oracle.Restatement().render(cfg, scene.desc, cam2, W, H, seeds)
    _assert_same(oracle, ostate, oimg, r.read_state(), r.read_framebuffer(), "orbit camera")
    r.close()


def test_empty_and_degenerate_inputs(prt, oracle):
    # no OBJ at all: empty-leaf root; 1x1 image; zero frames
    text = ('{"scene":{"spheres":[{"pos":[0,3,0],"radius":0.5,"material":{"color":[5,5,5],"type":0}}],'
            '"quads":[{"vertices":[0,0,0,4,0,0,0,0,4],"material":{"color":[1,1,1]}}]}}')
    scene = prt.HostScene(text, text=True)
    cfg = scene.config()
    for (W, H) in ((1, 1), (17, 3)):
        cam = prt.default_camera(max(W, 2), max(H, 2))
        r = prt.Renderer(cfg, device=0)
        r.upload_scene(scene)
        r.set_camera(cam)
        r.resize(W, H)
        r.render_frames(np.zeros(0, np.int32))
        assert (r.read_state()["samples"] == 0).all()
        seeds = prt.seed_pairs(30)
        r.render_frames(seeds)
        ostate, oimg = oracle.Restatement().render(cfg, scene.desc, cam, W, H, seeds)
        _assert_same(oracle, ostate, oimg, r.read_state(), r.read_framebuffer(), "no-OBJ scene %dx%d" % (W, H))
        r.close()


def test_errors_are_reported_not_fatal(prt):
    scene = prt.HostScene("cornell_diffuse.json")
    cfg = scene.config()
    r = prt.Renderer(cfg, device=0)
    with pytest.raises(prt.PrtError):
        r.render_frames(prt.seed_pairs(1))          # nothing uploaded yet
    r.upload_scene(scene)
    with pytest.raises(prt.PrtError):
        r.resize(0, 10)
    bad = prt.Config.from_buffer_copy(bytes(cfg))
    bad.abi_version = 99
    with pytest.raises(prt.PrtError):
        prt.Renderer(bad, device=0)
    with pytest.raises(prt.PrtError):
        prt.Renderer(cfg, device=4096)
    endless = prt.Config.from_buffer_copy(bytes(cfg))
    endless.marching_steps = 2 ** 31 - 1
    with pytest.raises(prt.PrtError, match="MARCHING_STEPS"):
        prt.Renderer(endless, device=0)
    r.close()
    # box primitives: geometry/box.cl is never included by the reference's kernel, they cannot render there either
    import json
    doc = json.load(open(os.path.join(prt.SCENES_DIR, "cornell_diffuse.json")))
    doc["scene"]["boxes"] = [{"pos": [0, 0, 0], "scale": [1, 1, 1], "material": {"type": 1, "color": [1, 1, 1]}}]
    boxed = prt.HostScene(json.dumps(doc), text=True)
    with pytest.raises(prt.PrtError, match="box"):
        prt.Renderer(boxed.config(), device=0)


def test_failed_reallocation_leaves_the_context_not_ready(prt, oracle):
    """a resize that cannot be satisfied must leave the context NOT READY (never "ready" with freed planes: the next
    launch would fault the GPU), and a later valid resize must bring it back; a rejected scene upload leaves the old
    scene in place"""
    g = np.load(os.path.join(GOLDEN, "cornell_diffuse.npz"))
    W, H, frames = int(g["width"]), int(g["height"]), int(g["frames"])
    scene, cfg, cam, env, r = _setup(prt, "cornell_diffuse", W, H)
    seeds = prt.seed_pairs(frames)
    with pytest.raises(prt.PrtError):
        r.resize(1 << 20, 1 << 20)                  # 6 planes x 16 TiB
    with pytest.raises(prt.PrtError, match="no frame size"):
        r.render_frames(seeds)                      # PRT_ERR_NOT_READY, no launch
    with pytest.raises(prt.PrtError):
        r.read_framebuffer()
    r.resize(W, H)
    # a scene the packer refuses (light index out of range) must not disturb the uploaded one
    bad_cfg = prt.Config.from_buffer_copy(bytes(cfg))
    bad_cfg.light_indices[0] = 1000
    r_bad = prt.Renderer(bad_cfg, device=0)
    with pytest.raises(prt.PrtError, match="light index"):
        r_bad.upload_scene(scene)
    with pytest.raises(prt.PrtError, match="no scene"):
        r_bad.render_frames(seeds)
    r_bad.close()
    desc = prt.SceneDesc.from_buffer_copy(bytes(scene.desc))
    desc.object_count[7] = desc.object_count[7] + 1          # counts no longer add up
    with pytest.raises(prt.PrtError, match="add up"):
        r.upload_scene(desc)
    r.render_frames(seeds)
    gstate = np.ascontiguousarray(g["state"]).view(oracle.PATH_STATE_DTYPE).reshape(-1)
    _assert_same(oracle, gstate, g["image"], r.read_state(), r.read_framebuffer(), "after failed resize / upload")
    r.close()


@pytest.mark.parametrize("streams", ["1", "2"])
def test_spp_frame_budget_exhausted_then_render_again(prt, oracle, streams, monkeypatch):
    """prt_render_spp that runs out of max_frames reports PRT_ERR_NOT_READY with every internal stream joined and the
    launch counters clean: the same context must render correctly afterwards"""
    monkeypatch.setenv("PRT_STREAMS", streams)
    monkeypatch.setenv("PRT_FRAMES_PER_LAUNCH", "8")
    gs = np.load(os.path.join(GOLDEN, "cornell_diffuse_spp.npz"))
    W, H, maxf, spp = int(gs["width"]), int(gs["height"]), int(gs["frames"]), int(gs["spp"])
    scene, cfg, cam, env, r = _setup(prt, "cornell_diffuse", W, H)
    with pytest.raises(prt.PrtError, match="max_frames"):
        r.render_spp(spp, prt.seed_pairs(12))       # 12 frames cannot finish 6 paths per pixel
    st = r.read_state()
    assert (st["samples"] >= 1).all() and not (st["samples"] == spp).all()
    # lanes run ahead of their launch while their wave waits (8-frame launches here), but never past the seed table: what
    # the caller finds is every pixel after exactly 12 frames of the "N spp" rule
    ostate, oimg = oracle.Restatement().render(cfg, scene.desc, cam, W, H, prt.seed_pairs(12), env=env, spp_limit=spp)
    _assert_same(oracle, ostate, oimg, st, r.read_framebuffer(), "exhausted spp call, streams=%s" % streams)
    r.reset()
    used = r.render_spp(spp, prt.seed_pairs(maxf))
    assert 0 < used <= maxf
    sstate = np.ascontiguousarray(gs["state"]).view(oracle.PATH_STATE_DTYPE).reshape(-1)
    _assert_same(oracle, sstate, gs["image"], r.read_state(), r.read_framebuffer(), "spp after an exhausted call, streams=%s" % streams)
    r.close()


@pytest.mark.parametrize("streams", ["1", "2", "3", "4"])
def test_internal_streams_are_invisible(prt, oracle, streams, monkeypatch):
    """the megakernel renders interleaved sets of tiles on PRT_STREAMS internal streams (default 2); any number gives
    the bits of the reference golden, in frame mode and in spp mode, on the context's own and on a caller's stream"""
    monkeypatch.setenv("PRT_STREAMS", streams)
    variant = "cornell_diffuse"
    g = np.load(os.path.join(GOLDEN, variant + ".npz"))
    W, H, frames = int(g["width"]), int(g["height"]), int(g["frames"])
    scene, cfg, cam, env, r = _setup(prt, variant, W, H)
    r.render_frames(prt.seed_pairs(frames))
    gstate = np.ascontiguousarray(g["state"]).view(oracle.PATH_STATE_DTYPE).reshape(-1)
    _assert_same(oracle, gstate, g["image"], r.read_state(), r.read_framebuffer(), "streams=%s vs golden" % streams)
    assert r.stats().concurrent == int(streams)
    gs = np.load(os.path.join(GOLDEN, "cornell_diffuse_spp.npz"))
    r.reset()
    used = r.render_spp(int(gs["spp"]), prt.seed_pairs(int(gs["frames"])))
    sstate = np.ascontiguousarray(gs["state"]).view(oracle.PATH_STATE_DTYPE).reshape(-1)
    _assert_same(oracle, sstate, gs["image"], r.read_state(), r.read_framebuffer(), "streams=%s spp vs golden" % streams)
    assert 0 < used <= int(gs["frames"])
    st = r.stats()
    assert st.kernel_ms > 0 and st.kernel_sum_ms >= 0.5 * st.kernel_ms and st.launches >= int(streams)
    r.close()

def test_run_ahead_is_invisible_at_full_size(prt, oracle, monkeypatch):
    """prt_render_spp lets a lane that waits for its wave start on the next launch's frames (the lead is kept per pixel in the
    state between launches).  BASELINE config 2's frame at 1920x1080, 24 spp in launches of 32 frames: with and without it
    (PRT_RUN_AHEAD) the path state, the framebuffer and the work counters are the same bits -- and the lead is gone at the end."""
    W, H, spp = 1920, 1080, 24
    monkeypatch.setenv("PRT_FRAMES_PER_LAUNCH", "32")
    out = []
    for flag in ("0", "1"):
        monkeypatch.setenv("PRT_RUN_AHEAD", flag)
        scene, cfg, cam, env, r = _setup(prt, "cornell_diffuse", W, H)
        used = r.render_spp(spp, prt.seed_pairs(spp * 16 + 64))
        c = r.counts(spp)
        out.append((r.read_state(), r.read_framebuffer(), c.segments, c.samples, c.finished_pixels, used, r.stats().launches))
        r.close()
    (s0, i0, seg0, smp0, fin0, used0, l0), (s1, i1, seg1, smp1, fin1, used1, l1) = out
    _assert_same(oracle, s0, i0, s1, i1, "run-ahead on vs off")
    assert (seg0, smp0, fin0) == (seg1, smp1, fin1) and fin1 == W * H and smp1 == spp * W * H
    assert used1 <= used0 and l1 <= l0           # pixels that ran ahead finish in no more launches

@pytest.mark.parametrize("fixture", list(VIEW_VARIANTS))
def test_debug_views_match_reference_golden(prt, oracle, fixture):
    """prt_config::view_option = VIEW_NORMAL / VIEW_BVH_HIT (kernels/main.cl:6-15): fixtures from the reference built with the view
    switched on; frame batches and an SDF scene against the oracle; the views the reference cannot render are refused"""
    base, view = VIEW_VARIANTS[fixture]
    g = np.load(os.path.join(GOLDEN, fixture + ".npz"))
    W, H, frames = int(g["width"]), int(g["height"]), int(g["frames"])
    scene_json, phase, use_env = VARIANTS[base]
    scene = prt.HostScene(scene_json)
    cfg = scene.config()
    cfg.phase_function = phase
    cfg.view_option = view
    env = prt.make_sky(64, 32) if use_env else None
    r = prt.Renderer(cfg, device=0)
    r.upload_scene(scene)
    if env is not None:
        r.upload_envmap(env)
    r.set_camera(variant_camera(prt, base, W, H))
    r.resize(W, H)
    seeds = prt.seed_pairs(frames)
    r.render_frames(seeds[:2 * 40])
    r.render_frames(seeds[2 * 40:], first_frame=41)
    gstate = np.ascontiguousarray(g["state"]).view(oracle.PATH_STATE_DTYPE).reshape(-1)
    _assert_same(oracle, gstate, g["image"], r.read_state(), r.read_framebuffer(), fixture)
    r.close()
    scene = prt.HostScene("cornell_sdf.json")
    cfg = scene.config()
    cfg.view_option = view
    cam = variant_camera(prt, "cornell_sdf", 29, 19)
    seeds = prt.seed_pairs(40)
    ostate, oimg = oracle.Restatement().render(cfg, scene.desc, cam, 29, 19, seeds, env=env)
    r = prt.Renderer(cfg, device=0)
    r.upload_scene(scene)
    if env is not None:
        r.upload_envmap(env)
    r.set_camera(cam)
    r.resize(29, 19)
    r.render_frames(seeds)
    _assert_same(oracle, ostate, oimg, r.read_state(), r.read_framebuffer(), "cornell_sdf view %d" % view)
    r.close()
    for bad in (2, 4, 8, 3):                                      # VIEW_STACK_INDEX, VIEW_ALBEDO, VIEW_SPECULAR, a mixture
        cfg.view_option = bad
        rb = prt.Renderer(cfg, device=0)
        with pytest.raises(prt.PrtError, match="view_option"):
            rb.upload_scene(scene)
        rb.close()


def _golden_through(prt, oracle, variant, what, **options):
    """renders a golden's inputs under the given prt_set_option choices; returns what the launches ran"""
    g = np.load(os.path.join(GOLDEN, variant + ".npz"))
    W, H, frames = int(g["width"]), int(g["height"]), int(g["frames"])
    scene, cfg, cam, env, r = _setup(prt, variant, W, H)
    for k, v in options.items():
        r.set_option(k, v)
    r.render_frames(prt.seed_pairs(frames))
    gstate = np.ascontiguousarray(g["state"]).view(oracle.PATH_STATE_DTYPE).reshape(-1)
    _assert_same(oracle, gstate, g["image"], r.read_state(), r.read_framebuffer(), "%s %s vs golden" % (variant, what))
    ran = r.kernel_variant()
    r.close()
    return ran


@pytest.mark.parametrize("scatter", [0, 1])
def test_pixel_to_wave_mappings_match_golden(prt, oracle, scatter):
    """a wave renders one 8x8 tile, or (launches with few rounds of waves) 64 pixels of 64 tiles spread over the launch: both
    mappings forced here (prt_set_option; what ran is read back), on a ragged frame (edge tiles: lanes outside the frame idle)
    and in spp mode"""
    mapping = "pixels=scattered" if scatter else "pixels=tiles"
    for variant in ("cornell_mixed", "cornell_sdf"):
        assert mapping in _golden_through(prt, oracle, variant, mapping, scatter=scatter)
    variant = "cornell_mixed"
    W2, H2, frames2 = 61, 43, 20                                  # neither a multiple of 8
    scene, cfg, cam2, env, r = _setup(prt, variant, W2, H2)
    r.set_option("scatter", scatter)
    seeds = prt.seed_pairs(frames2)
    ostate, oimg = oracle.Restatement().render(cfg, scene.desc, cam2, W2, H2, seeds, env=env)
    r.render_frames(seeds)
    assert mapping in r.kernel_variant()
    _assert_same(oracle, ostate, oimg, r.read_state(), r.read_framebuffer(), "%s ragged frame" % mapping)
    r.close()
    gs = np.load(os.path.join(GOLDEN, "cornell_diffuse_spp.npz"))
    scene, cfg, cam, env, r = _setup(prt, "cornell_diffuse", int(gs["width"]), int(gs["height"]))
    r.set_option("scatter", scatter)
    r.render_spp(int(gs["spp"]), prt.seed_pairs(int(gs["frames"])))
    assert mapping in r.kernel_variant()
    sstate = np.ascontiguousarray(gs["state"]).view(oracle.PATH_STATE_DTYPE).reshape(-1)
    _assert_same(oracle, sstate, gs["image"], r.read_state(), r.read_framebuffer(), "%s spp golden" % mapping)
    r.close()


@pytest.mark.parametrize("scatter", [0, 1])
@pytest.mark.parametrize("variant", list(VARIANTS))
def test_ray_pool_kernel_matches_golden(prt, oracle, variant, scatter):
    """render_kernel_rp (pt_pool.h, prt_set_option "pool"): shading waves that take only the step at the root of a walk and post the rays
    that go deeper into a pool in LDS, walker waves that walk the rays of the whole workgroup -- who walks a ray, next to which other
    rays and when is anybody's guess, the answer is not.  Every golden, both pixel mappings, what ran read back."""
    ran = _golden_through(prt, oracle, variant, "ray pool, scatter=%d" % scatter, pool=1, scatter=scatter)
    assert "pool" in ran and ("pixels=scattered" if scatter else "pixels=tiles") in ran, ran


def test_ray_pool_kernel_ragged_frame_and_spp_mode(prt, oracle):
    variant = "cornell_mixed"
    W2, H2, frames2 = 61, 43, 20                                  # neither a multiple of 8; 48 tiles = 10 workgroups of 5 shading waves, the last one short
    for scatter in (0, 1):
        scene, cfg, cam2, env, r = _setup(prt, variant, W2, H2)
        r.set_option("pool", 1)
        r.set_option("scatter", scatter)
        seeds = prt.seed_pairs(frames2)
        ostate, oimg = oracle.Restatement().render(cfg, scene.desc, cam2, W2, H2, seeds, env=env)
        r.render_frames(seeds)
        assert "pool" in r.kernel_variant()
        _assert_same(oracle, ostate, oimg, r.read_state(), r.read_framebuffer(), "ray pool, ragged frame, scatter=%d" % scatter)
        r.close()
    gs = np.load(os.path.join(GOLDEN, "cornell_diffuse_spp.npz"))
    for per_launch in (0, 7):                                     # one launch; several (leads of the run-ahead carried from launch to launch)
        scene, cfg, cam, env, r = _setup(prt, "cornell_diffuse", int(gs["width"]), int(gs["height"]))
        r.set_option("pool", 1)
        r.set_option("frames_per_launch", per_launch)
        r.render_spp(int(gs["spp"]), prt.seed_pairs(int(gs["frames"])))
        assert "pool" in r.kernel_variant()
        sstate = np.ascontiguousarray(gs["state"]).view(oracle.PATH_STATE_DTYPE).reshape(-1)
        _assert_same(oracle, sstate, gs["image"], r.read_state(), r.read_framebuffer(), "ray pool, spp golden, %d frames per launch" % per_launch)
        r.close()


def test_ray_pool_kernel_through_the_big_tree(prt, oracle, monkeypatch):
    """the 871 k-triangle stand-in (22 stack levels, nearly every ray goes past the root; launches that take their tiles in the
    launcher's order and report what they cost): the ray pool's render equals render_kernel's bit for bit, in samples-per-pixel mode"""
    W, H, spp = 2560, 1440, 2
    monkeypatch.setenv("PRT_FRAMES_PER_LAUNCH", "8")
    prt.ensure_dragon_standin()
    seeds = prt.seed_pairs(spp * 16 + 64)
    scene = prt.HostScene("cornell_dragon.json")
    r = prt.Renderer(scene.config(), device=0)
    r.upload_scene(scene)
    r.set_camera(prt.default_camera(W, H))
    r.resize(W, H)
    out = []
    for pool in (0, 1):
        r.set_option("pool", pool)
        r.reset()
        r.render_spp(spp, seeds)
        ran = r.kernel_variant()
        assert ("pool" in ran) == bool(pool) and "expensive first" in ran, ran
        c = r.counts(spp)
        out.append((r.read_state(), r.read_framebuffer(), c.segments, c.samples, c.finished_pixels))
    r.close()
    _assert_same(oracle, out[0][0], out[0][1], out[1][0], out[1][1], "ray pool vs render_kernel through the big tree")
    assert out[1][2:] == out[0][2:] and out[1][4] == W * H


@pytest.mark.parametrize("per_wave", [32, 16])
@pytest.mark.parametrize("scatter", [0, 1])
def test_fewer_pixels_per_wave_match_golden(prt, oracle, per_wave, scatter):
    """FrameArgs::sub_shift (prt_set_option "pix_per_wave"): a wave renders 32 or 16 pixels and 2 or 4 waves share a tile (or, scattered, the
    launch's pixels) -- what launches that leave wave slots empty take by themselves (one rank's share of a frame split N ways).  Pixels are
    independent, so this is bit-exact by construction; checked on goldens, on a ragged frame against the oracle and in samples-per-pixel
    mode, both pixel mappings, what ran read back."""
    tag = "%d per wave" % per_wave
    for variant in ("cornell_mixed", "cornell_media_hg", "cornell_diffuse"):
        ran = _golden_through(prt, oracle, variant, tag, pix_per_wave=per_wave, scatter=scatter)
        assert tag in ran and ("pixels=scattered" if scatter else "pixels=tiles") in ran, ran
    W2, H2, frames2 = 61, 43, 20                                  # neither a multiple of 8
    scene, cfg, cam2, env, r = _setup(prt, "cornell_mixed", W2, H2)
    r.set_option("pix_per_wave", per_wave)
    r.set_option("scatter", scatter)
    seeds = prt.seed_pairs(frames2)
    ostate, oimg = oracle.Restatement().render(cfg, scene.desc, cam2, W2, H2, seeds, env=env)
    r.render_frames(seeds)
    assert tag in r.kernel_variant()
    _assert_same(oracle, ostate, oimg, r.read_state(), r.read_framebuffer(), "%s, ragged frame, scatter=%d" % (tag, scatter))
    r.close()
    gs = np.load(os.path.join(GOLDEN, "cornell_diffuse_spp.npz"))
    scene, cfg, cam, env, r = _setup(prt, "cornell_diffuse", int(gs["width"]), int(gs["height"]))
    r.set_option("pix_per_wave", per_wave)
    r.set_option("scatter", scatter)
    r.set_option("frames_per_launch", 7)
    r.render_spp(int(gs["spp"]), prt.seed_pairs(int(gs["frames"])))
    assert tag in r.kernel_variant()
    sstate = np.ascontiguousarray(gs["state"]).view(oracle.PATH_STATE_DTYPE).reshape(-1)
    _assert_same(oracle, sstate, gs["image"], r.read_state(), r.read_framebuffer(), "%s, spp golden" % tag)
    r.close()


@pytest.mark.parametrize("waves", [5, 6])
@pytest.mark.parametrize("scatter", [0, 1])
@pytest.mark.parametrize("variant", ["cornell_diffuse", "cornell_media_hg", "cornell_sdf", "cornell_mixed", "cornell_coat", "cornell_roughdiel"])
def test_wave_count_builds_match_golden(prt, oracle, variant, waves, scatter):
    """every kernel variant is built for 5 and 6 waves per SIMD (96 / 80 registers); the launcher picks one per launch (6 in
    launches of whole tiles, 5 in small launches with scattered pixels).  Both builds x both pixel mappings, forced and read back"""
    ran = _golden_through(prt, oracle, variant, "waves=%d scatter=%d" % (waves, scatter), waves=waves, scatter=scatter)
    assert "waves=%d" % waves in ran and ("pixels=scattered" if scatter else "pixels=tiles") in ran, ran


COMPILED_SETS = {"cornell_diffuse": "<LIGHT|DIFF>", "cornell_media": "<LIGHT|DIFF,medium>", "cornell_coat": "<LIGHT|DIFF|COAT; Beckmann>",
                 "cornell_quadlight": "<LIGHT|DIFF|COAT>", "cornell_roughcond": "<LIGHT|DIFF|ROUGH_COND; GGX>",
                 "cornell_roughdiel": "<LIGHT|DIFF|DIEL|ROUGH_DIEL; GGX>"}


@pytest.mark.parametrize("variant", list(VARIANTS))
def test_compiled_material_sets_and_generic_dispatch_match_golden(prt, oracle, variant):
    """the reference compiles exactly the scene's ACTIVE_MATS (include/CL/cl_kernel.h:226-345); here the sets of the BASELINE
    configs are compiled and every other scene runs the variant that dispatches at run time.  Every golden goes through the
    instance the launcher picks AND through the generic one (prt_set_option "generic"); what ran is read back."""
    picked = _golden_through(prt, oracle, variant, "scene's own set")
    if variant in COMPILED_SETS:
        assert COMPILED_SETS[variant] in picked, picked
        if ";" in COMPILED_SETS[variant]:                        # ... and through the set's instance with every microfacet distribution
            wide = _golden_through(prt, oracle, variant, "set with all distributions", any_dist=1)
            assert COMPILED_SETS[variant].split(";")[0] + ">" in wide, wide
    forced = _golden_through(prt, oracle, variant, "generic set", generic=1)
    assert "generic" in forced, forced


def _mean_and_error(img):
    """per channel: mean over the pixels and its Monte-Carlo error from the spread of 8 x 8 block means"""
    h, w = img.shape[0] // 8 * 8, img.shape[1] // 8 * 8
    a = np.nan_to_num(img[:h, :w, :3].astype(np.float64), nan=0.0, posinf=0.0, neginf=0.0)
    blocks = a.reshape(h // 8, 8, w // 8, 8, 3).mean(axis=(1, 3)).reshape(-1, 3)
    return a.mean(axis=(0, 1)), blocks.std(axis=0, ddof=1) / np.sqrt(blocks.shape[0])


def test_environment_importance_sampling_converges_to_the_lookup_render(prt, oracle):
    """prt_config::env_importance_sampling (north_star; NOT in the reference, which only looks the map up where a ray escapes): off by
    default.  On: the expectation of every pixel must be the default mode's -- an open box under a sky with a sun, both modes converged,
    image means within the Monte-Carlo error of their difference -- the noise must be lower (that is what it is for), and the device
    code must equal its host compilation bit for bit (tests/emu: there is no oracle for a mode the reference does not have)."""
    W, H, frames = 128, 96, 8192
    scene = prt.HostScene("cornell_open.json")
    cam = prt.default_camera(W, H)
    env = prt.make_sky(256, 128)
    res = {}
    for mode in (0, 1):
        cfg = scene.config()
        cfg.env_importance_sampling = mode
        imgs = []
        for first in (1, 1 + frames):                                   # two independent renders per mode
            r = prt.Renderer(cfg, device=0)
            r.upload_scene(scene)
            r.upload_envmap(env)
            r.set_camera(cam)
            r.resize(W, H)
            r.render_frames(prt.seed_pairs(frames, first_frame=first))
            imgs.append(r.read_framebuffer().astype(np.float64))
            ran = r.kernel_variant()
            r.close()
        assert ("env_importance_sampling" in ran) == bool(mode), ran
        m, e = _mean_and_error(0.5 * (imgs[0] + imgs[1]))
        d = np.nan_to_num(imgs[0][..., :3] - imgs[1][..., :3])
        res[mode] = (m, e, np.sqrt((d ** 2).mean()) / m.mean())
    (m0, e0, n0), (m1, e1, n1) = res[0], res[1]
    z = (m1 - m0) / np.sqrt(e0 ** 2 + e1 ** 2)
    assert (np.abs(z) < 4.5).all() and (np.abs(m1 / m0 - 1.0) < 0.01).all(), "means %s vs %s (z = %s)" % (m0, m1, z)
    assert n1 < 0.9 * n0, "noise with importance sampling %.4f, without %.4f" % (n1, n0)
    # device == host compilation of the same code, bit for bit
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "emu"))
    import emu_api
    W2, H2, f2 = 29, 19, 96
    cfg = scene.config()
    cfg.env_importance_sampling = 1
    cam2 = prt.default_camera(W2, H2)
    seeds = prt.seed_pairs(f2)
    small = prt.make_sky(64, 32)
    hstate, himg = emu_api.render(oracle.PATH_STATE_DTYPE, cfg, scene.desc, cam2, W2, H2, seeds, env=small)
    r = prt.Renderer(cfg, device=0)
    r.upload_scene(scene)
    r.upload_envmap(small)
    r.set_camera(cam2)
    r.resize(W2, H2)
    r.render_frames(seeds)
    _assert_same(oracle, hstate, himg, r.read_state(), r.read_framebuffer(), "env importance sampling: GPU vs host compilation")
    r.close()
    # refused where it is not built
    bad = prt.HostScene("cornell_media.json")
    cfgm = bad.config()
    cfgm.env_importance_sampling = 1
    rb = prt.Renderer(cfgm, device=0)
    with pytest.raises(prt.PrtError, match="env_importance_sampling"):
        rb.upload_scene(bad)
    rb.close()


def test_expensive_tiles_first_is_invisible(prt, oracle, monkeypatch):
    """Through a big tree prt_render_spp sorts a sub-part's tiles by what their waves reported in its first launch (iterations) and starts
    the expensive ones first from then on (FrameArgs::tile_order, the ORDER build of the kernel; kept until scene, camera or frame
    change).  The 871 k-triangle stand-in at 2560x1440 (the smallest 16:9 frame whose sub-parts are not scattered), 3 spp in launches of 8
    frames: with the order (measured in the first render, reused in the second) and without it the state and the image are the same bits."""
    W, H, spp = 2560, 1440, 3
    monkeypatch.setenv("PRT_FRAMES_PER_LAUNCH", "8")
    prt.ensure_dragon_standin()
    seeds = prt.seed_pairs(spp * 16 + 64)
    scene = prt.HostScene("cornell_dragon.json")
    r = prt.Renderer(scene.config(), device=0)
    r.upload_scene(scene)
    r.set_camera(prt.default_camera(W, H))
    r.resize(W, H)
    out = []
    # render 0 measures the order in its first launch (8 frames) and uses it from its second on; render 1 is ONE launch per sub-part
    # (the default launch length) and sets the option again to the same value, which must keep the measured order: its launch 0 starts
    # with it, or nothing would say "expensive first"; render 2 runs without
    for order, per_launch in ((1, 8), (1, 0), (0, 8)):
        r.set_option("tile_order", order)
        r.set_option("frames_per_launch", per_launch)
        r.reset()
        r.render_spp(spp, seeds)
        assert ("expensive first" in r.kernel_variant()) == bool(order), r.kernel_variant()
        if per_launch == 0:
            assert r.stats().launches == 2, r.stats().launches
        c = r.counts(spp)
        out.append((r.read_state(), r.read_framebuffer(), c.segments, c.samples, c.finished_pixels))
    r.close()
    for k in (1, 2):
        _assert_same(oracle, out[0][0], out[0][1], out[k][0], out[k][1], "tile order: render %d vs the first" % k)
        assert out[k][2:] == out[0][2:] and out[k][4] == W * H


def test_pacing_is_invisible(prt, oracle):
    """prt_render_spp with "pace" (default on): from the second launch on a pixel whose own mean path length exceeds the frame's owes the
    launches proportionally more frames, so that the pixels with long paths are not left for the tail of the render.  A pixel's frame numbers
    and seeds are its own, so this is a schedule like any other: many short launches (the pace engages from the second one), samples-per-pixel
    mode, rough dielectric (the config whose per-pixel path lengths differ most) -- with and without, the same bits, and fewer launches with."""
    W, H, spp = 320, 200, 24
    scene, cfg, cam, env, r = _setup(prt, "cornell_roughdiel", W, H)
    seeds = prt.seed_pairs(spp * 40 + 64)
    out = []
    for pace in (0, 1):
        r.set_option("pace", pace)
        r.set_option("frames_per_launch", 16)
        r.reset()
        r.render_spp(spp, seeds)
        c = r.counts(spp)
        out.append((r.read_state(), r.read_framebuffer(), c.segments, c.samples, c.finished_pixels, r.stats().launches))
    r.close()
    _assert_same(oracle, out[0][0], out[0][1], out[1][0], out[1][1], "pace on vs off")
    assert out[0][2:5] == out[1][2:5] and out[0][4] == W * H
    assert out[1][5] < out[0][5], (out[0][5], out[1][5])          # the paced render needs fewer launches: nobody is left for a tail


def test_a_launch_that_does_not_report_aborts_the_render(prt):
    """prt_render_spp arms a pinned word with ~0 before every launch of a sub-part and the last wave of the launch overwrites it with the
    number of unfinished pixels (render_kernel).  A report that never arrives (faked: option "test_drop_report" points the kernel at a spare
    word) must end the call with PRT_ERR_HIP through abort_streams -- not be read as 1.8e19 pixels and run on to "max_frames reached" --
    and leave the state marked unusable until prt_reset."""
    W, H, spp = 640, 640, 2                                  # 6 400 tiles: two sub-parts
    seeds = prt.seed_pairs(spp * 16 + 64)
    scene = prt.HostScene("cornell_diffuse.json")
    r = prt.Renderer(scene.config(), device=0)
    r.upload_scene(scene)
    r.set_camera(prt.default_camera(W, H))
    r.resize(W, H)
    r.set_option("test_drop_report", 1)
    with pytest.raises(prt.PrtError) as ei:
        r.render_spp(spp, seeds)
    assert ei.value.code == prt.PRT_ERR_HIP and "without reporting" in str(ei.value), str(ei.value)
    r.set_option("test_drop_report", 0)
    with pytest.raises(prt.PrtError) as ei:                  # the state may hold run-ahead leads: unusable until reset
        r.render_spp(spp, seeds)
    assert ei.value.code == prt.PRT_ERR_NOT_READY
    r.reset()
    r.render_spp(spp, seeds)
    assert r.counts(spp).finished_pixels == W * H
    r.close()


@pytest.mark.parametrize("case", ["config2", "config3a", "config4"])
def test_hip_agrees_with_the_reference_under_glibc_math(prt, case):
    """Math this repository did not write, ON THE GPU PATH.  Every bit-exact test here is relative to include/prt_detmath.h (it feeds the
    reference build's OpenCL runtime stand-in, the CPU restatement and the HIP kernels alike).  tests/golden/libm_means.npz holds BASELINE
    configs 2 / 3a / 4 rendered by the reference's own kernel text with the GNU C library behind its scalar transcendental built-ins
    (generator: tests/golden/make_libm_means.py; 128 x 128 pixels x 4 096 frames).  The two libraries part ways pixel by pixel at the first
    decision an ulp flips (a path tracer is chaotic), so what must agree is the estimate: per channel, the image mean of the HIP render of
    the same inputs within 0.1 % of the reference's and within 4.5 Monte-Carlo errors of the difference (from the spread of the 8 x 8 block
    means: blocks decorrelate the pixels that share RNG streams)."""
    g = np.load(os.path.join(GOLDEN, "libm_means.npz"))
    size, frames = int(g["size"]), int(g["frames"])
    scene_json, phase, use_env = {"config2": ("cornell_diffuse.json", 0, False), "config3a": ("cornell_roughcond.json", 0, True),
                                  "config4": ("cornell_media.json", 1, True)}[case]
    scene = prt.HostScene(scene_json)
    cfg = scene.config()
    cfg.phase_function = phase
    r = prt.Renderer(cfg, device=0)
    r.upload_scene(scene)
    if use_env:
        r.upload_envmap(prt.make_sky(64, 32))
    r.set_camera(prt.default_camera(size, size))
    r.resize(size, size)
    r.render_frames(prt.seed_pairs(frames))
    img = r.read_framebuffer().reshape(size, size, 4)[..., :3].astype(np.float64)
    assert r.counts(0).segments == int(g[case + "_segments"])          # (one segment per pixel and frame, whatever the math)
    r.close()
    ok = np.isfinite(img).all(axis=2)
    a = np.where(ok[..., None], img, 0.0)
    h = size // 8
    cnt = ok.reshape(h, 8, h, 8).sum(axis=(1, 3))
    mine = a.reshape(h, 8, h, 8, 3).sum(axis=(1, 3)) / np.maximum(cnt, 1)[..., None]
    ref, ref_cnt = g[case + "_blocks"], g[case + "_counts"]
    use = (cnt >= 16) & (ref_cnt >= 16)                                 # blocks with enough finite pixels in both renders
    assert use.mean() > 0.9, use.mean()
    for ch in range(3):
        m_hip, m_ref = mine[..., ch][use].mean(), ref[..., ch][use].mean()
        d = (mine[..., ch] - ref[..., ch])[use]
        sigma = d.std(ddof=1) / np.sqrt(d.size)
        z = (m_hip - m_ref) / sigma
        assert abs(m_hip / m_ref - 1.0) < 1e-3 and abs(z) < 4.5, (case, "RGB"[ch], m_hip, m_ref, m_hip / m_ref, z)


def test_options_are_validated(prt):
    scene = prt.HostScene("cornell_diffuse.json")
    r = prt.Renderer(scene.config(), device=0)
    assert r.kernel_variant() == ""
    for name, value in (("waves", 4), ("waves", 7), ("scatter", 2), ("generic", 2), ("any_dist", 2), ("tri_q", 17), ("frames_per_launch", -1), ("tile_order", 2), ("test_drop_report", 2), ("pix_per_wave", 48), ("pool", 2), ("pace", 2), ("no_such_option", 1)):
        with pytest.raises(prt.PrtError):
            r.set_option(name, value)
    r.close()


def test_fast_reciprocal_is_the_ieee_divide_on_every_float(prt):
    """hw_recip (estimate + one fma Newton step, divide for the extreme exponents) replaces the 1/x of the slab and
    triangle tests; it must be the correctly rounded reciprocal for all 2^32 inputs"""
    scene = prt.HostScene("cornell_diffuse.json")
    r = prt.Renderer(scene.config(), device=0)
    z = np.zeros(65536, dtype=np.float32)
    bad = r.selftest_math(17, z, z)
    r.close()
    assert bad.sum() == 0, "hw_recip differs from 1.0f/x on %d inputs" % int(bad.sum())


def test_fast_square_root_is_the_ieee_one_on_every_float(prt):
    """hw_sqrt (hardware 1/sqrt estimate, x * r, one fma correction step; the IEEE expansion outside 2^-100 .. 2^100) replaces the
    square roots of the device code; it must be the correctly rounded square root for all 2^32 inputs"""
    scene = prt.HostScene("cornell_diffuse.json")
    r = prt.Renderer(scene.config(), device=0)
    z = np.zeros(65536, dtype=np.float32)
    bad = r.selftest_math(19, z, z)
    r.close()
    assert bad.sum() == 0, "hw_sqrt differs from sqrtf(x) on %d inputs" % int(bad.sum())


@pytest.mark.parametrize("c", [1.0, 3.0, 0.7531, 16777215.0, 1.1920929e-07 * 3, 2.0 ** 40, 2.0 ** -40, 2.0 ** 41, 1e-20, 0.0, 5.960465e-08])
def test_quad_edge_predicate_is_the_divide_on_every_float(prt, c):
    """hit_quad compares x / c with 0 and 1 without dividing (pt_device.h out_of_unit_range); the predicate must equal
    the reference expression for every binary32 x, for divisors in and out of the fast range"""
    scene = prt.HostScene("cornell_diffuse.json")
    r = prt.Renderer(scene.config(), device=0)
    z = np.zeros(65536, dtype=np.float32)
    bad = r.selftest_math(18, z, np.full(65536, c, dtype=np.float32))
    r.close()
    assert bad.sum() == 0, "c=%g: predicate differs from the divide on %d inputs" % (c, int(bad.sum()))


def _chain_bvh(levels):
    """a hand-made BVH whose walk can stack `levels` entries: every chain node has two inner children, a
    two-leaf stub and the rest of the chain (reference node layout, include/BVH/bvh.h:24-30)"""
    node_t = np.dtype([("bounds", "<f4", 6), ("first", "<u4"), ("count", "<u4"), ("leaf", "u1"), ("_p", "u1", 3)])
    n_nodes = 1 + 4 * levels + 2
    nodes = np.zeros(n_nodes, dtype=node_t)
    nodes["bounds"] = np.array([-1, 1, -1, 1, -1, 1], dtype=np.float32)
    tri = 0
    at, nxt = 0, 1
    for _ in range(levels):
        stub, chain = nxt, nxt + 1                      # children of the chain node sit side by side
        nodes[at]["first"], nodes[at]["count"], nodes[at]["leaf"] = stub, 0, 0
        l0 = nxt + 2
        nodes[stub]["first"], nodes[stub]["leaf"] = l0, 0
        for leaf in (l0, l0 + 1):
            nodes[leaf]["first"], nodes[leaf]["count"], nodes[leaf]["leaf"] = tri, 1, 1
            tri += 1
        at, nxt = chain, nxt + 4
    nodes[at]["first"], nodes[at]["leaf"] = nxt, 0       # the chain ends in one more two-leaf node
    for leaf in (nxt, nxt + 1):
        nodes[leaf]["first"], nodes[leaf]["count"], nodes[leaf]["leaf"] = tri, 1, 1
        tri += 1
    assert nxt + 2 == n_nodes
    verts = np.zeros((tri * 3, 4), dtype=np.float32)
    verts[1::3, 0] = 0.01
    verts[2::3, 1] = 0.01
    normals = np.zeros_like(verts)
    normals[:, 2] = 1.0
    return nodes, verts, normals, np.arange(tri, dtype=np.uint64)


def test_callers_tree_with_shared_and_fat_leaves(prt, oracle):
    """a tree the builder would never make but the reference kernel accepts: loose boxes, a 40-triangle leaf, sibling leaves that share
    triangles (one slot per reference after packing; pending runs of 110 triangles)"""
    from test_emu import callers_tree
    scene, desc, keep = callers_tree(prt, oracle)
    W, H, frames = 40, 24, 16
    cfg = scene.config()
    cam = prt.default_camera(W, H)
    seeds = prt.seed_pairs(frames)
    ostate, oimg = oracle.Restatement().render(cfg, desc, cam, W, H, seeds, threads=8)
    r = prt.Renderer(cfg, device=0)
    r.upload_scene(desc)
    r.set_camera(cam)
    r.resize(W, H)
    r.render_frames(seeds)
    _assert_same(oracle, ostate, oimg, r.read_state(), r.read_framebuffer(), "caller's tree")
    r.close()


@pytest.mark.parametrize("levels,ok", [(64, True), (65, False)])
def test_traversal_stack_limit_is_the_references(prt, levels, ok):
    """the reference's closest-hit stack has 64 entries (bvh.cl:131) and overflows silently beyond; here a tree
    that could stack more is refused at upload, one that fits renders"""
    import ctypes as C
    scene = prt.HostScene("cornell_diffuse.json")
    nodes, verts, normals, idx = _chain_bvh(levels)
    desc = prt.SceneDesc.from_buffer_copy(bytes(scene.desc))
    desc.vertices = verts.ctypes.data_as(C.c_void_p)
    desc.normals = normals.ctypes.data_as(C.c_void_p)
    desc.primitive_indices = idx.ctypes.data_as(C.c_void_p)
    desc.triangle_count = len(idx)
    desc.bvh_nodes = nodes.ctypes.data_as(C.c_void_p)
    desc.bvh_node_count = len(nodes)
    r = prt.Renderer(scene.config(), device=0)
    if ok:
        r.upload_scene(desc)
        r.set_camera(prt.default_camera(16, 8))
        r.resize(16, 8)
        r.render_frames(prt.seed_pairs(4))
        assert np.isfinite(r.read_framebuffer()).all()
    else:
        with pytest.raises(prt.PrtError, match="64 traversal-stack entries"):
            r.upload_scene(desc)
    r.close()


@pytest.mark.parametrize("damage", ["child_out_of_range", "cycle", "leaf_range", "primitive_index"])
def test_malformed_bvh_is_refused_at_upload(prt, damage):
    """the kernels index the node / triangle buffers unchecked, so prt_upload_scene is where a broken tree must stop"""
    import ctypes as C
    scene = prt.HostScene("cornell_diffuse.json")
    nodes, verts, normals, idx = _chain_bvh(4)
    if damage == "child_out_of_range":
        nodes[0]["first"] = len(nodes) - 1
    elif damage == "cycle":
        inner = [i for i in range(1, len(nodes)) if not nodes[i]["leaf"]]
        nodes[inner[-1]]["first"] = 0                      # points back at the root's... node 0 and 1
    elif damage == "leaf_range":
        leaf = [i for i in range(len(nodes)) if nodes[i]["leaf"]][-1]
        nodes[leaf]["count"] = 1000
    elif damage == "primitive_index":
        idx[3] = 10 ** 6
    desc = prt.SceneDesc.from_buffer_copy(bytes(scene.desc))
    desc.vertices = verts.ctypes.data_as(C.c_void_p)
    desc.normals = normals.ctypes.data_as(C.c_void_p)
    desc.primitive_indices = idx.ctypes.data_as(C.c_void_p)
    desc.triangle_count = len(idx)
    desc.bvh_nodes = nodes.ctypes.data_as(C.c_void_p)
    desc.bvh_node_count = len(nodes)
    r = prt.Renderer(scene.config(), device=0)
    with pytest.raises(prt.PrtError):
        r.upload_scene(desc)
    r.close()


def test_dragon_standin_matches_oracle(prt, oracle):
    """871 k triangles, BVH depth 23: deep stacks (LDS + scratch levels), MALL-resident geometry"""
    prt.ensure_dragon_standin()
    W, H, frames = 128, 72, 48
    scene = prt.HostScene("cornell_dragon.json")
    cfg, cam, seeds = scene.config(), prt.default_camera(W, H), prt.seed_pairs(frames)
    r = prt.Renderer(cfg, device=0)
    r.upload_scene(scene)
    r.set_camera(cam)
    r.resize(W, H)
    r.render_frames(seeds)
    ostate, oimg = oracle.Restatement().render(cfg, scene.desc, cam, W, H, seeds, threads=16)
    _assert_same(oracle, ostate, oimg, r.read_state(), r.read_framebuffer(), "dragon stand-in")
    r.close()


@pytest.mark.parametrize("variant", list(VARIANTS))
@pytest.mark.parametrize("min_lanes", [1, 64])
def test_schedule_independence(prt, oracle, variant, min_lanes):
    """the lane machine lets pixels drift in frame number; WHEN a wave ends its walk phases (prt_set_walk_min_lanes:
    1 = every walk runs to its end, the lock-step schedule; 64 = a phase ends as soon as one lane is done, the most
    drift) must not change a bit: same state and image as the reference golden, and a render continued under the
    other schedule equals one rendered in one go"""
    g = np.load(os.path.join(GOLDEN, variant + ".npz"))
    W, H, frames = int(g["width"]), int(g["height"]), int(g["frames"])
    scene, cfg, cam, env, r = _setup(prt, variant, W, H)
    r.set_walk_min_lanes(min_lanes)
    r.render_frames(prt.seed_pairs(frames))
    gstate = np.ascontiguousarray(g["state"]).view(oracle.PATH_STATE_DTYPE).reshape(-1)
    _assert_same(oracle, gstate, g["image"], r.read_state(), r.read_framebuffer(), "%s min_lanes %d vs golden" % (variant, min_lanes))
    more = prt.seed_pairs(frames + 20)[2 * frames:]
    r.set_walk_min_lanes(65 - min_lanes)
    r.render_frames(more, first_frame=frames + 1)
    s_mix, i_mix = r.read_state(), r.read_framebuffer()
    r.close()
    scene, cfg, cam, env, r2 = _setup(prt, variant, W, H)
    r2.render_frames(prt.seed_pairs(frames + 20))
    _assert_same(oracle, r2.read_state().view(oracle.PATH_STATE_DTYPE), r2.read_framebuffer(), s_mix, i_mix, variant + " mixed schedules vs default")
    r2.close()


def test_spp_mode_lockstep_schedule(prt, oracle):
    g = np.load(os.path.join(GOLDEN, "cornell_diffuse_spp.npz"))
    W, H, maxf, spp = int(g["width"]), int(g["height"]), int(g["frames"]), int(g["spp"])
    scene, cfg, cam, env, r = _setup(prt, "cornell_diffuse", W, H)
    r.set_walk_min_lanes(1)
    r.render_spp(spp, prt.seed_pairs(maxf))
    gstate = np.ascontiguousarray(g["state"]).view(oracle.PATH_STATE_DTYPE).reshape(-1)
    _assert_same(oracle, gstate, g["image"], r.read_state(), r.read_framebuffer(), "lock-step spp golden")
    r.close()


def test_full_size_frame_against_oracle_strips_and_properties(prt, oracle):
    """BASELINE config 2 at its real size (1920x1080): the oracle renders three 4-row strips of the frame
    (global pixel coordinates) and must match the GPU's full-frame pixels bit for bit; plus the
    size-independent properties: interleaved row blocks == full frame, frame batches compose, counters add up."""
    W, H, frames = 1920, 1080, 40
    scene, cfg, cam, env, r = _setup(prt, "cornell_diffuse", W, H)
    seeds = prt.seed_pairs(frames)
    r.render_frames(seeds[:2 * 17])
    r.render_frames(seeds[2 * 17:], first_frame=18)                       # two batches
    img = r.read_framebuffer()
    state = r.read_state().reshape(H, W)
    st = r.counts()
    assert st.segments == frames * W * H                                   # every pixel advances one segment per frame (Q19)
    assert st.samples == int(state["samples"].sum()) and int(state["acc"][..., 3].sum()) == st.segments
    r.close()
    rs = oracle.Restatement()
    for row0 in (0, 538, 1076):
        ostate, oimg = rs.render(cfg, scene.desc, cam, W, H, seeds, row0=row0, rows=4, threads=16)
        _assert_same(oracle, ostate, oimg, state[row0:row0 + 4].reshape(-1), img[row0:row0 + 4], "full-size strip at row %d" % row0)
    # the multi-GPU partition at full size: 8 parts, check two of them
    for part in (0, 5):
        rows = np.array([y for y in range(H) if (y // 16) % 8 == part])
        rt = prt.Renderer(cfg, device=0)
        rt.upload_scene(scene)
        rt.set_camera(cam)
        rt.set_row_blocks(W, H, 16, 8, part)
        rt.render_frames(seeds)
        _assert_same(oracle, state[rows].reshape(-1).view(oracle.PATH_STATE_DTYPE), img[rows], rt.read_state(), rt.read_framebuffer(),
                     "full-size row blocks, part %d of 8" % part)
        rt.close()


@pytest.mark.parametrize("variant", ["cornell_roughcond", "cornell_roughdiel", "cornell_media_hg"])
def test_full_size_configs_3_and_4_against_oracle_strips(prt, oracle, variant):
    """BASELINE configs 3a / 3b / 4 at their real size: 1920x1080 with the 1024x512 environment map (the small-size tests
    use a 64x32 one).  Three 4-row oracle strips (global pixel coordinates: top, middle, bottom of the frame) must match
    the GPU's full-frame pixels bit for bit, the counters must add up, and two frame batches must compose."""
    W, H, frames = 1920, 1080, 28
    scene_json, phase, use_env = VARIANTS[variant]
    scene = prt.HostScene(scene_json)
    cfg = scene.config()
    cfg.phase_function = phase
    cam = prt.default_camera(W, H)
    env = prt.make_sky(1024, 512)
    r = prt.Renderer(cfg, device=0)
    r.upload_scene(scene)
    r.upload_envmap(env)
    r.set_camera(cam)
    r.resize(W, H)
    seeds = prt.seed_pairs(frames)
    r.render_frames(seeds[:2 * 11])
    r.render_frames(seeds[2 * 11:], first_frame=12)
    img = r.read_framebuffer()
    state = r.read_state().reshape(H, W)
    st = r.counts()
    assert st.segments == frames * W * H
    assert st.samples == int(state["samples"].sum()) and int(state["acc"][..., 3].sum()) == st.segments
    r.close()
    rs = oracle.Restatement()
    for row0 in (0, 538, 1076):
        ostate, oimg = rs.render(cfg, scene.desc, cam, W, H, seeds, env=env, row0=row0, rows=4, threads=16)
        _assert_same(oracle, ostate, oimg, state[row0:row0 + 4].reshape(-1), img[row0:row0 + 4], "%s full-size strip at row %d" % (variant, row0))


def test_full_size_config_5_against_oracle_strips(prt, oracle):
    """BASELINE config 5 at its real frame size: the 871 k-triangle stand-in at 3840x2160 (8.3 M pixels: tile indices past
    2^17, pixel indices past 2^23), a few dozen frames.  Oracle strips at the top, through the mesh and at the bottom;
    one of the eight row-block parts of the multi-GPU split against the same frame."""
    prt.ensure_dragon_standin()
    W, H, frames = 3840, 2160, 24
    scene = prt.HostScene("cornell_dragon.json")
    cfg = scene.config()
    cam = prt.default_camera(W, H)
    r = prt.Renderer(cfg, device=0)
    r.upload_scene(scene)
    r.set_camera(cam)
    r.resize(W, H)
    seeds = prt.seed_pairs(frames)
    r.render_frames(seeds)
    img = r.read_framebuffer()
    state = r.read_state().reshape(H, W)
    st = r.counts()
    assert st.segments == frames * W * H
    assert st.samples == int(state["samples"].sum()) and int(state["acc"][..., 3].sum()) == st.segments
    r.close()
    rs = oracle.Restatement()
    for row0 in (0, 1300, 2156):
        ostate, oimg = rs.render(cfg, scene.desc, cam, W, H, seeds, row0=row0, rows=4, threads=16)
        _assert_same(oracle, ostate, oimg, state[row0:row0 + 4].reshape(-1), img[row0:row0 + 4], "dragon 4K strip at row %d" % row0)
    part = 3
    rows = np.array([y for y in range(H) if (y // 16) % 8 == part])
    rt = prt.Renderer(cfg, device=0)
    rt.upload_scene(scene)
    rt.set_camera(cam)
    rt.set_row_blocks(W, H, 16, 8, part)
    rt.render_frames(seeds)
    _assert_same(oracle, state[rows].reshape(-1).view(oracle.PATH_STATE_DTYPE), img[rows], rt.read_state(), rt.read_framebuffer(),
                 "dragon 4K row blocks, part %d of 8" % part)
    rt.close()


def test_full_size_spp_render_is_complete(prt, oracle):
    """1920x1080 at 8 spp to completion: every pixel froze at exactly 8 paths, alpha = segments / samples"""
    W, H, spp = 1920, 1080, 8
    scene, cfg, cam, env, r = _setup(prt, "cornell_diffuse", W, H)
    r.render_spp(spp, prt.seed_pairs(spp * cfg.max_bounces + 64))
    st = r.counts(spp)
    assert st.finished_pixels == W * H and st.samples == spp * W * H
    img = r.read_framebuffer()
    state = r.read_state()
    assert np.isfinite(img).all() and (img[..., :3] >= 0).all()
    assert np.array_equal(img[..., 3].reshape(-1), state["acc"][:, 3] / np.float32(spp))
    r.close()


def test_axis_parallel_rays_nan_slabs(prt, oracle):
    """centre column of a pinhole camera at odd width: dir.x == 0, origin.x == 0 -> inf / NaN slabs; the device's
    v_min_f32 / v_max_f32 must treat the NaN like the oracle's fmin / fmax"""
    W, H, frames = 33, 21, 80
    scene, cfg, cam, env, r = _setup(prt, "cornell_coat", W, H)
    cam = prt.orbit_camera(W, H, d_aperture=-1.0)
    r.set_camera(cam)
    seeds = prt.seed_pairs(frames)
    r.render_frames(seeds)
    ostate, oimg = oracle.Restatement().render(cfg, scene.desc, cam, W, H, seeds, threads=8)
    _assert_same(oracle, ostate, oimg, r.read_state(), r.read_framebuffer(), "axis-parallel rays")
    r.close()


def test_tonemap_matches_the_reference_shader_and_cli_writes_png(prt, oracle, tmp_path):
    """display side (row N3): shaders/tonemapper.glsl evaluated in float64 numpy vs prt_tonemap_rgba8 (+-1 LSB),
    and the headless CLI (the main.cpp equivalent) end to end: render -> tonemap -> PNG"""
    import struct
    import subprocess
    import zlib
    W, H = 96, 64
    scene, cfg, cam, env, r = _setup(prt, "cornell_coat", W, H)
    r.render_spp(4, prt.seed_pairs(4 * cfg.max_bounces + 64))
    fb = r.read_framebuffer().astype(np.float64)
    ldr = r.tonemap_rgba8()
    r.close()
    yy, xx = np.mgrid[0:H, 0:W]
    px, py = 1 - 2 * (xx + 0.5) / W, 1 - 2 * (yy + 0.5) / H
    vig = (1.25 / (1.1 + 1.1 * (px * px + py * py))) ** 2
    sm = lambda e0, e1, x: (lambda t: t * t * (3 - 2 * t))(np.clip((x - e0) / (e1 - e0), 0, 1))
    vig = 0.75 + 0.25 * sm(0.1, 1.1, vig)
    curve = lambda x: (57.25 * x * x) / (57.25 * x * x + x + 56.25)
    col = curve(fb[..., :3] * vig[..., None]) / curve(1.2)
    col = np.clip(sm(-0.025, 1.0, col) ** (1 / 2.2), 0, 1)
    want = np.rint(col * 255)
    assert np.abs(ldr[..., :3].astype(np.float64) - want).max() <= 1 and (ldr[..., 3] == 255).all()
    # CLI
    exe = os.path.join(os.path.dirname(prt.__file__), "prt_render")
    out = tmp_path / "render.png"
    res = subprocess.run([exe, "-scene", os.path.join(prt.SCENES_DIR, "cornell_coat.json"), "-models", prt.MODELS_DIR + "/",
                          "-width", str(W), "-height", str(H), "-spp", "4", "-out", str(out)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert res.returncode == 0, res.stdout
    raw = out.read_bytes()
    assert raw[:8] == b"\x89PNG\r\n\x1a\n"
    pos, idat, dims = 8, b"", None
    while pos < len(raw):
        n, typ = struct.unpack(">I4s", raw[pos:pos + 8])
        data = raw[pos + 8:pos + 8 + n]
        assert struct.unpack(">I", raw[pos + 8 + n:pos + 12 + n])[0] == (zlib.crc32(typ + data) & 0xffffffff)
        if typ == b"IHDR":
            dims = struct.unpack(">II", data[:8])
        if typ == b"IDAT":
            idat += data
        pos += 12 + n
    assert dims == (W, H)
    pix = np.frombuffer(zlib.decompress(idat), dtype=np.uint8).reshape(H, W * 4 + 1)
    assert (pix[:, 0] == 0).all()
    png = pix[:, 1:].reshape(H, W, 4)
    assert np.array_equal(png[::-1], ldr)             # the PNG is top-down, the framebuffer bottom-up; same render (deterministic)
    # -encoder 1: render.hdr in the working directory, the LINEAR picture as Radiance RGBE (saveImage, include/GL/cl_gl_interop.h:151-156)
    res = subprocess.run([exe, "-scene", os.path.join(prt.SCENES_DIR, "cornell_coat.json"), "-models", prt.MODELS_DIR + "/",
                          "-width", str(W), "-height", str(H), "-spp", "4", "-encoder", "1"], cwd=str(tmp_path),
                         stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert res.returncode == 0, res.stdout
    hdr = prt.load_hdr(str(tmp_path / "render.hdr"))
    lin = fb[::-1, :, :3]
    m = lin.max(axis=2)
    ok = m >= 1e-32
    assert hdr.shape == lin.shape and (hdr[~ok] == 0).all()
    assert (np.abs(hdr[ok] - lin[ok]).max(axis=1) <= m[ok] / 127.0).all()      # 8-bit mantissas against the shared exponent


def test_device_functions_match_reference_kat(prt):
    """prt_selftest_fn on the GPU against the per-function known-answer vectors of the reference build (tests/test_kat.py)"""
    from test_kat import assert_kat_equal, kat_tables
    scene = prt.HostScene("cornell_diffuse.json")
    r = prt.Renderer(scene.config(), device=0)
    n = 0
    for name, fn, params, cases, expect, cols in kat_tables():
        assert_kat_equal(name, r.selftest_fn(fn, params, cases), expect, cols)
        n += 1
    assert n >= 40
    from test_kat import check_env_lookup
    check_env_lookup(r.selftest_fn)             # read_imagef semantics worked out from the OpenCL 1.2 specification
    r.close()


def test_bench_line_keeps_the_contract(prt):
    """`python bench.py` as the driver runs it (fewer spp): ONE JSON line with the contract's keys, the roofline and the CPU baseline"""
    import json
    import subprocess
    import sys
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1", "--spp", "16"],
                         capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.split("\n") if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    j = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
                "data", "config", "roofline", "cpu_baseline"):
        assert key in j, key
    assert j["unit"] == "Msamples/s" and j["n_gpus"] == 1 and j["steps"] == 2 and j["warmup"] == 1 and j["higher_is_better"] is True
    assert j["value"] > 0 and j["ms_per_step"] > 0 and j["vs_baseline"] is None and j["dtype"] == "f32" and j["data"] == "synthetic"
    assert "workload" in j["config"] and "model" not in j["config"]
    r = j["roofline"]
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic", "valu", "kernel"):
        assert key in r, key
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-4
    assert "render_kernel<LIGHT|DIFF>" in r["kernel"]
    c = j["cpu_baseline"]
    for key in ("value", "unit", "cores", "kind", "sample"):
        assert key in c, key
    assert c["kind"] in ("port", "reference") and c["value"] > 0 and c["cores"] >= 1
