"""CPU tests of the product's DEVICE code: csrc/hip/pt_device.h (the lane machine, BVH steps, BSDFs, media ...)
compiled for the host and run by the wave emulator of tests/emu/pt_emu.cpp, against
  * the golden fixtures produced by the reference build (tests/golden/*.npz), and
  * oracle/pt_oracle.c on ragged sizes,
bit for bit, under the kernel's default schedule and under RANDOM walk-phase lengths (lanes drift in frame number:
which lanes walk together must not change a bit).  No GPU involved: code generation for gfx950 is what -m gpu checks."""
import os
import sys

import numpy as np
import pytest

from conftest import ALPHA_VARIANTS, variant_config, GOLDEN, ROOT, VARIANTS, VIEW_VARIANTS, variant_camera

sys.path.insert(0, os.path.join(ROOT, "tests", "emu"))


@pytest.fixture(scope="module")
def emu():
    import emu_api
    emu_api.lib()
    return emu_api


def _scene(prt, variant, W, H):
    scene_json, phase, use_env = VARIANTS[variant]
    scene = prt.HostScene(scene_json)
    cfg = variant_config(scene, variant)
    cfg.phase_function = phase
    return scene, cfg, variant_camera(prt, variant, W, H), (prt.make_sky(64, 32) if use_env else None)


def _same(oracle, s0, i0, s1, i1, what):
    bad = oracle.state_fields_equal(s0, s1.view(oracle.PATH_STATE_DTYPE))
    assert not bad, "%s: path state differs in %s" % (what, bad)
    assert oracle.images_equal(i0, i1), "%s: framebuffer differs" % what


@pytest.mark.parametrize("variant", list(VARIANTS))
@pytest.mark.parametrize("sched", [(8, 0), (1, 0), (8, 12345)])
def test_device_code_on_host_matches_reference_golden(prt, oracle, emu, variant, sched):
    g = np.load(os.path.join(GOLDEN, variant + ".npz"))
    W, H, frames = int(g["width"]), int(g["height"]), int(g["frames"])
    scene, cfg, cam, env = _scene(prt, variant, W, H)
    state, img = emu.render(oracle.PATH_STATE_DTYPE, cfg, scene.desc, cam, W, H, prt.seed_pairs(frames), env=env,
                            walk_min_lanes=sched[0], sched_seed=sched[1])
    gstate = np.ascontiguousarray(g["state"]).view(oracle.PATH_STATE_DTYPE).reshape(-1)
    _same(oracle, gstate, g["image"], state, img, "%s schedule %s" % (variant, sched))


@pytest.mark.parametrize("variant", ["cornell_diffuse", "cornell_coat", "cornell_roughcond", "cornell_roughdiel", "cornell_media"])
def test_generic_material_dispatch_on_host_matches_reference_golden(prt, oracle, emu, variant, monkeypatch):
    """scenes whose ACTIVE_MATS is one of the compiled sets, through the run-time dispatch instead (LaunchOpts::generic)"""
    monkeypatch.setenv("PT_EMU_GENERIC", "1")
    g = np.load(os.path.join(GOLDEN, variant + ".npz"))
    W, H, frames = int(g["width"]), int(g["height"]), int(g["frames"])
    scene, cfg, cam, env = _scene(prt, variant, W, H)
    state, img = emu.render(oracle.PATH_STATE_DTYPE, cfg, scene.desc, cam, W, H, prt.seed_pairs(frames), env=env, sched_seed=4711)
    gstate = np.ascontiguousarray(g["state"]).view(oracle.PATH_STATE_DTYPE).reshape(-1)
    _same(oracle, gstate, g["image"], state, img, "%s generic set" % variant)


@pytest.mark.parametrize("fixture", list(VIEW_VARIANTS))
def test_debug_views_on_host_match_reference_golden(prt, oracle, emu, fixture):
    """the PT_MATS_VIEW kernel variants (prt_config::view_option = VIEW_NORMAL / VIEW_BVH_HIT) under a random schedule; plus an SDF
    scene against the oracle (a hit that comes out of the hit cache must show what the reference's fresh walk would leave)"""
    base, view = VIEW_VARIANTS[fixture]
    g = np.load(os.path.join(GOLDEN, fixture + ".npz"))
    W, H, frames = int(g["width"]), int(g["height"]), int(g["frames"])
    scene, cfg, cam, env = _scene(prt, base, W, H)
    cfg.view_option = view
    state, img = emu.render(oracle.PATH_STATE_DTYPE, cfg, scene.desc, cam, W, H, prt.seed_pairs(frames), env=env, sched_seed=77)
    gstate = np.ascontiguousarray(g["state"]).view(oracle.PATH_STATE_DTYPE).reshape(-1)
    _same(oracle, gstate, g["image"], state, img, fixture)
    scene, cfg, cam, env = _scene(prt, "cornell_sdf", 29, 19)
    cfg.view_option = view
    seeds = prt.seed_pairs(40)
    ostate, oimg = oracle.Restatement().render(cfg, scene.desc, cam, 29, 19, seeds, env=env, threads=4)
    state, img = emu.render(oracle.PATH_STATE_DTYPE, cfg, scene.desc, cam, 29, 19, seeds, env=env)
    _same(oracle, ostate, oimg, state, img, "cornell_sdf view %d" % view)


def test_spp_mode_and_batches_on_host(prt, oracle, emu):
    g = np.load(os.path.join(GOLDEN, "cornell_diffuse_spp.npz"))
    W, H, maxf, spp = int(g["width"]), int(g["height"]), int(g["frames"]), int(g["spp"])
    scene, cfg, cam, env = _scene(prt, "cornell_diffuse", W, H)
    seeds = prt.seed_pairs(maxf)
    # the "N spp" rule in launches of 32 frames, as prt_render_spp issues them, under a random schedule
    state = img = None
    for f in range(0, maxf, 32):
        state, img = emu.render(oracle.PATH_STATE_DTYPE, cfg, scene.desc, cam, W, H, seeds[2 * f:2 * (f + 32)], first_frame=1 + f,
                                state=state, img=img, spp_limit=spp, sched_seed=99 + f)
    gstate = np.ascontiguousarray(g["state"]).view(oracle.PATH_STATE_DTYPE).reshape(-1)
    _same(oracle, gstate, g["image"], state, img, "spp golden")
    # the same with FrameArgs::run_ahead, as prt_render_spp launches: a lane that has done the 32 frames of a launch goes on into
    # the frames of the next ones while its wave waits for other lanes -- pixels end the launches at different frame numbers and
    # still reach the same final state
    state = img = None
    ahead = np.zeros(W * H, dtype=np.uint32)
    ran_ahead = 0
    for f in range(0, maxf, 32):
        state, img = emu.render(oracle.PATH_STATE_DTYPE, cfg, scene.desc, cam, W, H, seeds[2 * f:], first_frame=1 + f,
                                state=state, img=img, spp_limit=spp, sched_seed=7 + f, walk_min_lanes=6, window=32, ahead=ahead)
        ran_ahead += int((ahead > 0).sum())
        assert int(ahead.max()) <= maxf - min(f + 32, maxf)
    assert ran_ahead > 0 and not ahead.any()
    _same(oracle, gstate, g["image"], state, img, "spp golden, lanes running ahead")


@pytest.mark.parametrize("variant", ["cornell_mixed", "cornell_media_hg", "cornell_sdf"])
def test_ragged_frame_and_row_blocks_on_host(prt, oracle, emu, variant):
    W, H, frames = 45, 27, 64
    scene, cfg, cam, env = _scene(prt, variant, W, H)
    seeds = prt.seed_pairs(frames)
    ostate, oimg = oracle.Restatement().render(cfg, scene.desc, cam, W, H, seeds, env=env, threads=8)
    state, img = emu.render(oracle.PATH_STATE_DTYPE, cfg, scene.desc, cam, W, H, seeds, env=env, sched_seed=7)
    _same(oracle, ostate, oimg, state, img, variant + " ragged")
    blocks = (4, 3, 1)
    bstate, bimg = emu.render(oracle.PATH_STATE_DTYPE, cfg, scene.desc, cam, W, H, seeds, env=env, sched_seed=8, blocks=blocks)
    rows = [r for r in range(H) if (r // blocks[0]) % blocks[1] == blocks[2]]
    assert oracle.images_equal(oimg[rows], bimg), "row blocks differ from the full frame"


def _mean_and_error(img):
    """per channel: mean over the finite pixels and its Monte-Carlo error from the spread of 4 x 4 block means"""
    h, w = img.shape[0] // 4 * 4, img.shape[1] // 4 * 4
    a = np.nan_to_num(img[:h, :w, :3].astype(np.float64), nan=0.0, posinf=0.0, neginf=0.0)
    blocks = a.reshape(h // 4, 4, w // 4, 4, 3).mean(axis=(1, 3)).reshape(-1, 3)
    return a.mean(axis=(0, 1)), blocks.std(axis=0, ddof=1) / np.sqrt(blocks.shape[0])


def test_environment_importance_sampling_has_the_same_expectation_on_host(prt, oracle, emu):
    """prt_config::env_importance_sampling (not in the reference): the light-sample strategy of a vertex is a coin flip between the light
    and the environment map (sampled by luminance x sin theta, power heuristic against the BSDF sample).  The estimator changes, the
    expectation must not: an open box under the sky, both modes, image means within the Monte-Carlo error of their difference"""
    W, H, frames = 24, 16, 3072
    scene = prt.HostScene("cornell_open.json")
    cam = prt.default_camera(W, H)
    env = prt.make_sky(64, 32)
    seeds = prt.seed_pairs(frames)
    out = {}
    for mode in (0, 1):
        cfg = scene.config()
        cfg.env_importance_sampling = mode
        _, img = emu.render(oracle.PATH_STATE_DTYPE, cfg, scene.desc, cam, W, H, seeds, env=env)
        out[mode] = _mean_and_error(img)
    (m0, e0), (m1, e1) = out[0], out[1]
    z = (m1 - m0) / np.sqrt(e0 ** 2 + e1 ** 2)
    assert (np.abs(z) < 4.5).all() and (np.abs(m1 / m0 - 1.0) < 0.05).all(), "means %s vs %s (z = %s)" % (m0, m1, z)


def callers_tree(prt, oracle):
    """a CALLER'S tree, legal for the reference kernel but unlike anything the builder makes: loose boxes (the whole scene), a leaf of 40
    triangles, two sibling leaves that SHARE triangles (ranges 20..59 and 30..99), 100 of the mesh's triangles referenced in all"""
    import ctypes as C
    scene = prt.HostScene("cornell_coat.json")
    node_t = np.dtype([("bounds", "<f4", 6), ("first", "<u4"), ("count", "<u4"), ("leaf", "u1"), ("_p", "u1", 3)])
    nodes = np.zeros(5, dtype=node_t)
    nodes["bounds"] = np.array([-3, 3, -1, 5, -3, 3], dtype=np.float32)
    nodes[0]["first"], nodes[0]["leaf"] = 1, 0                                   # root: children 1, 2
    nodes[1]["first"], nodes[1]["count"], nodes[1]["leaf"] = 0, 40, 1            # a fat leaf
    nodes[2]["first"], nodes[2]["leaf"] = 3, 0                                   # inner: children 3, 4
    nodes[3]["first"], nodes[3]["count"], nodes[3]["leaf"] = 20, 40, 1
    nodes[4]["first"], nodes[4]["count"], nodes[4]["leaf"] = 30, 70, 1
    desc = prt.SceneDesc.from_buffer_copy(bytes(scene.desc))
    desc.bvh_nodes = nodes.ctypes.data_as(C.c_void_p)
    desc.bvh_node_count = len(nodes)
    return scene, desc, nodes


def test_callers_tree_with_shared_and_fat_leaves_on_host(prt, oracle, emu):
    scene, desc, keep = callers_tree(prt, oracle)
    W, H, frames = 24, 16, 12
    cfg = scene.config()
    cam = prt.default_camera(W, H)
    seeds = prt.seed_pairs(frames)
    ostate, oimg = oracle.Restatement().render(cfg, desc, cam, W, H, seeds, threads=4)
    for sched in ((8, 0), (8, 99)):
        state, img = emu.render(oracle.PATH_STATE_DTYPE, cfg, desc, cam, W, H, seeds, walk_min_lanes=sched[0], sched_seed=sched[1])
        _same(oracle, ostate, oimg, state, img, "caller's tree, schedule %s" % (sched,))
