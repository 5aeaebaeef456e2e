"""The CPU oracle (oracle/pt_oracle.c) against the golden fixtures produced by the reference
build, against the reference build itself where it is present (development container), and
through size-independent properties."""
import os

import numpy as np
import pytest

from conftest import ALPHA_VARIANTS, variant_config, GOLDEN, VARIANTS, VIEW_VARIANTS, variant_camera


def setup(prt, variant, W, H):
    scene_json, phase, use_env = VARIANTS[variant]
    scene = prt.HostScene(scene_json)
    cfg = variant_config(scene, variant)
    cfg.phase_function = phase
    return scene, cfg, variant_camera(prt, variant, W, H), (prt.make_sky(64, 32) if use_env else None)


@pytest.mark.parametrize("variant", list(VARIANTS) + ["cornell_diffuse_spp"])
def test_oracle_reproduces_reference_golden(prt, oracle, variant):
    g = np.load(os.path.join(GOLDEN, variant + ".npz"))
    W, H, frames, spp = int(g["width"]), int(g["height"]), int(g["frames"]), int(g["spp"])
    scene, cfg, cam, env = setup(prt, variant.replace("_spp", ""), W, H)
    state, img = oracle.Restatement().render(cfg, scene.desc, cam, W, H, prt.seed_pairs(frames), env=env, spp_limit=spp, threads=4)
    gstate = np.ascontiguousarray(g["state"]).view(oracle.PATH_STATE_DTYPE).reshape(-1)
    assert oracle.state_fields_equal(gstate, state) == []
    assert oracle.images_equal(g["image"], img)


@pytest.mark.parametrize("fixture", list(VIEW_VARIANTS))
def test_oracle_debug_views_reproduce_reference_golden(prt, oracle, fixture):
    """kernels/main.cl:6-15: VIEW_NORMAL / VIEW_BVH_HIT (the reference built with VIEW_OPTION switched, build_ref.py --view): every
    frame the accumulator is overwritten with ray.normal as radiance() leaves it, and the image is the accumulator"""
    base, view = VIEW_VARIANTS[fixture]
    g = np.load(os.path.join(GOLDEN, fixture + ".npz"))
    W, H, frames = int(g["width"]), int(g["height"]), int(g["frames"])
    assert int(g["view"]) == view
    scene, cfg, cam, env = setup(prt, base, W, H)
    cfg.view_option = view
    state, img = oracle.Restatement().render(cfg, scene.desc, cam, W, H, prt.seed_pairs(frames), env=env, threads=4)
    gstate = np.ascontiguousarray(g["state"]).view(oracle.PATH_STATE_DTYPE).reshape(-1)
    assert oracle.state_fields_equal(gstate, state) == []
    assert oracle.images_equal(g["image"], img)
    assert np.isnan(img).any() and (state["acc"][:, 3] == 1.0).all()      # rays that hit nothing leave normalize(0); alpha is always 1


@pytest.mark.parametrize("variant", list(VARIANTS))
def test_oracle_equals_reference_build(prt, oracle, variant):
    """different size / frame count / camera than the goldens, straight against oracle/_ref"""
    if not oracle.ref_available(variant):
        pytest.skip("oracle/_ref not built (only possible where /root/reference exists)")
    W, H, frames = 45, 27, 70
    scene, cfg, cam, env = setup(prt, variant, W, H)
    cam = prt.orbit_camera(W, H, d_yaw=-0.3, d_pitch=0.15, d_radius=-0.1)
    seeds = prt.seed_pairs(frames)
    rstate, rimg = oracle.RefOracle(variant).render(scene.desc, bytes(cam), W, H, seeds, env=env, threads=4)
    state, img = oracle.Restatement().render(cfg, scene.desc, cam, W, H, seeds, env=env, threads=4)
    assert oracle.state_fields_equal(rstate, state) == []
    assert oracle.images_equal(rimg, img)


def test_reference_shadow_stack_never_needed_more_than_eight(prt, oracle):
    """the reference's any-hit stack has 8 entries and no guard; the _ref build raises
    it to 64.  On the teapot scenes the deepest any-hit stack ever used stays below 8, so the patched
    and the original kernel are the same function on every fixture."""
    scene, cfg, cam, env = setup(prt, "cornell_coat", 64, 48)
    rs = oracle.Restatement()
    rs.render(cfg, scene.desc, cam, 64, 48, prt.seed_pairs(64), threads=4)
    max_stack, max_shadow = rs.last_diag
    assert 0 < max_shadow <= 8 and max_stack <= 64


def test_pixels_are_independent_tiles_and_row_blocks(prt, oracle):
    W, H, frames = 40, 37, 40
    scene, cfg, cam, env = setup(prt, "cornell_diffuse", W, H)
    seeds = prt.seed_pairs(frames)
    rs = oracle.Restatement()
    fs, fi = rs.render(cfg, scene.desc, cam, W, H, seeds, threads=4)
    fs = fs.reshape(H, W)
    s1, i1 = rs.render(cfg, scene.desc, cam, W, H, seeds, row0=10, rows=9, threads=2)
    assert oracle.state_fields_equal(fs[10:19].reshape(-1), s1) == [] and oracle.images_equal(fi[10:19], i1)
    for part in range(3):
        rows = np.array([y for y in range(H) if (y // 16) % 3 == part])
        s2, i2 = rs.render(cfg, scene.desc, cam, W, H, seeds, blocks=(16, 3, part), threads=2)
        assert oracle.state_fields_equal(fs[rows].reshape(-1), s2) == [] and oracle.images_equal(fi[rows], i2)


def test_frame_batches_compose(prt, oracle):
    W, H = 24, 16
    scene, cfg, cam, env = setup(prt, "cornell_coat", W, H)
    seeds = prt.seed_pairs(50)
    rs = oracle.Restatement()
    s_all, i_all = rs.render(cfg, scene.desc, cam, W, H, seeds, threads=2)
    s, _ = rs.render(cfg, scene.desc, cam, W, H, seeds[:2 * 20], threads=2)
    s, i = rs.render(cfg, scene.desc, cam, W, H, seeds[2 * 20:], first_frame=21, state=s, threads=2)
    assert oracle.state_fields_equal(s_all, s) == [] and oracle.images_equal(i_all, i)


def test_spp_rule_properties(prt, oracle):
    W, H, spp = 32, 20, 5
    scene, cfg, cam, env = setup(prt, "cornell_roughcond", W, H)
    seeds = prt.seed_pairs(spp * cfg.max_bounces + 64)
    state, img = oracle.Restatement().render(cfg, scene.desc, cam, W, H, seeds, spp_limit=spp, threads=4)
    assert (state["samples"] == spp).all() and (state["reset"] != 0).all()
    assert (state["acc"][:, 3] >= spp).all() and (state["acc"][:, 3] <= spp * (cfg.max_bounces + 1)).all()
    assert np.allclose(img[..., 3].reshape(-1), state["acc"][:, 3] / spp)              # alpha = segments / samples (Q19)
    assert np.isfinite(img).all() and (img[..., :3] >= 0).all()


def test_dragon_standin_oracle_equals_reference_build(prt, oracle):
    """config 5 geometry (871 k triangles, tree depth 23): needs any-hit stack depths > 8, where the
    unpatched reference is undefined; the reference build runs with the raised stack"""
    if not oracle.ref_available("cornell_diffuse"):
        pytest.skip("oracle/_ref not built")
    prt.ensure_dragon_standin()
    W, H, frames = 40, 24, 40
    scene = prt.HostScene("cornell_dragon.json")
    assert scene.desc.triangle_count == 871200
    cfg, cam, seeds = scene.config(), prt.default_camera(W, H), prt.seed_pairs(frames)
    rstate, rimg = oracle.RefOracle("cornell_diffuse").render(scene.desc, bytes(cam), W, H, seeds, threads=8)   # same specialisation: LIGHT|DIFF
    rs = oracle.Restatement()
    state, img = rs.render(cfg, scene.desc, cam, W, H, seeds, threads=8)
    assert oracle.state_fields_equal(rstate, state) == [] and oracle.images_equal(rimg, img)
    assert rs.last_diag[1] > 8          # the any-hit stack really goes deeper than the reference's 8 entries


def test_axis_parallel_rays_nan_slabs(prt, oracle):
    """pinhole camera + odd width: the centre column has dir.x == 0 and origin.x == 0 exactly, so the slab test
    (bvh.cl:4-9) produces inf and NaN (0 * inf) in every node -- fmin/fmax must ignore the NaN operand exactly as
    the reference build does"""
    if not oracle.ref_available("cornell_coat"):
        pytest.skip("oracle/_ref not built")
    W, H, frames = 33, 21, 60
    scene, cfg, _, env = setup(prt, "cornell_coat", W, H)
    cam = prt.orbit_camera(W, H, d_aperture=-1.0)                          # apertureRadius clamps to 0: pinhole
    assert cam.apertureRadius == 0.0 and cam.view[0] == 0.0 and cam.position[0] == 0.0
    seeds = prt.seed_pairs(frames)
    rstate, rimg = oracle.RefOracle("cornell_coat").render(scene.desc, bytes(cam), W, H, seeds, threads=4)
    state, img = oracle.Restatement().render(cfg, scene.desc, cam, W, H, seeds, threads=4)
    assert oracle.state_fields_equal(rstate, state) == [] and oracle.images_equal(rimg, img)
    # premise (by construction, camera.cl:26-52): view.x = up.x = position.x = 0 -> hAxis = (+-1,0,0), vAxis.x = 0,
    # and at x = (W-1)/2 the factor 2*sx-1 is exactly 0, so origin.x = dir.x = 0 for the whole centre column


LIBM_CASES = [("cornell_diffuse", "cornell_diffuse_libm", False, 0), ("cornell_roughcond", "cornell_roughcond_libm", True, 0),
              ("cornell_media_hg", "cornell_media_hg_libm", True, 1)]


@pytest.mark.parametrize("std,libm,use_env,phase", LIBM_CASES)
def test_reference_build_with_glibc_math_agrees_as_an_estimate(prt, oracle, std, libm, use_env, phase):
    """The reference's kernel text linked against TWO built-in libraries: include/prt_detmath.h (the stated one, shared by the restatement
    and the HIP kernels) and, for the scalar transcendentals, the GNU C library's libm (oracle/ref/build_ref.py --math libm) -- math this
    repository did not write.  Individual pixels part ways at the first decision an ulp flips; the ESTIMATE must not move: image means
    within the Monte-Carlo error of their difference (full-length run: tools/independent_math.py -> profiles/r03_independent_math.txt)."""
    if not (oracle.ref_available(std) and oracle.ref_available(libm)):
        pytest.skip("reference build not present (development container only)")
    size, frames = 48, 768
    scene = prt.HostScene(VARIANTS[std][0])
    cam = prt.default_camera(size, size)
    seeds = prt.seed_pairs(frames)
    env = prt.make_sky(64, 32) if use_env else None
    imgs = []
    for name in (std, libm):
        _, img = oracle.RefOracle(name).render(scene.desc, bytes(cam), size, size, seeds, env=env, threads=8)
        imgs.append(img.astype(np.float64)[..., :3])
    a, b = imgs
    finite = np.isfinite(a).all(axis=2) & np.isfinite(b).all(axis=2)
    assert finite.mean() > 0.8                                # (the reference leaves NaN in some pixels of the medium scenes: 0 * inf in a weight)
    same = ((a == b) | ~finite[..., None]).all(axis=2).mean()
    assert 0.0 < same < 1.0                                   # some pixels never met a flipped decision, most did
    d = np.where(finite[..., None], a - b, 0.0)
    blocks = d.reshape(size // 8, 8, size // 8, 8, 3).mean(axis=(1, 3)).reshape(-1, 3)
    sigma = blocks.std(axis=0, ddof=1) / np.sqrt(blocks.shape[0])
    mean = np.where(finite[..., None], a, 0.0).mean(axis=(0, 1))
    z = d.mean(axis=(0, 1)) / sigma
    assert (np.abs(z) < 4.5).all() and (np.abs(d.mean(axis=(0, 1)) / mean) < 0.02).all(), "z = %s, relative difference of the means %s" % (z, d.mean(axis=(0, 1)) / mean)
