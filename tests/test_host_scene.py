"""Host model (scene loader, camera, seed protocol, OBJ reader, BVH builder) against the buffers
the REFERENCE's own host code produced (tests/golden/*.sceneblob, written by oracle/ref/ref_host.cpp
around include/Scene/scene.h + src/Camera/camera.cpp)."""
import ctypes as C
import os
import struct

import numpy as np
import pytest

from conftest import ALPHA_VARIANTS, variant_config, GOLDEN, ROOT, VARIANTS


def read_blob(path):
    d = open(path, "rb").read()
    o, out = 0, {}
    while o < len(d):
        name = d[o:o + 16].split(b"\0")[0].decode()
        n = struct.unpack("<Q", d[o + 16:o + 24])[0]
        out[name] = d[o + 24:o + 24 + n]
        o += 24 + n
    return out


MESH_MASK = np.ones(256, bool)
MESH_MASK[56:64] = False      # Material padding
MESH_MASK[80:128] = False     # padding between pos and joker
MESH_MASK[193:] = False       # tail padding


@pytest.mark.parametrize("variant", list(VARIANTS))
def test_scene_loader_matches_reference_loader(prt, variant):
    b = read_blob(os.path.join(GOLDEN, variant + ".sceneblob"))
    scene = prt.HostScene(VARIANTS[variant][0])
    d = scene.desc
    n = d.object_count[7]
    assert list(d.object_count) == list(struct.unpack("<8I", b["counts"]))
    mine = np.frombuffer(C.string_at(d.meshes, n * 256), dtype=np.uint8).reshape(n, 256)
    ref = np.frombuffer(b["meshes"], dtype=np.uint8).reshape(n, 256)
    assert np.array_equal(mine[:, MESH_MASK], ref[:, MESH_MASK])
    assert C.string_at(d.obj_material, 56) == b["objmat"][:56]
    ints = struct.unpack("<16i", b["ints"])
    cfg = variant_config(scene, variant)
    assert cfg.alpha_testing == ints[15]
    assert (cfg.max_bounces, cfg.max_diff_bounces, cfg.max_spec_bounces, cfg.max_trans_bounces, cfg.max_scattering_events,
            cfg.marching_steps, cfg.shadow_marching_steps) == ints[:7]
    assert cfg.active_mats == ints[7]
    assert cfg.geom_flags == (1 * ints[8]) | (4 * ints[9]) | (2 * ints[10]) | (8 * ints[11])
    assert cfg.light_count == ints[12] and cfg.has_global_medium == ints[13]
    lights = np.frombuffer(b["lights"], dtype=np.uint32)
    assert list(cfg.light_indices)[:len(lights)] == list(lights)
    med = struct.unpack("<5f", b["medium"])
    if ints[13]:
        # the kernel text carries "%f" of these (include/CL/cl_kernel.h:72-108)
        rt = [np.float32(float("%f" % v)) for v in med[:4]]
        assert [cfg.fog_density, cfg.fog_sigma_a, cfg.fog_sigma_s, cfg.fog_sigma_t] == [float(x) for x in rt]
        assert cfg.fog_abs_only == int(med[4])


def test_default_camera_matches_reference_camera(prt):
    b = read_blob(os.path.join(GOLDEN, "cornell_coat.sceneblob"))     # built at 64x64
    cam = prt.default_camera(64, 64)
    assert bytes(cam)[:72] == b["camera"][:72]
    cam2 = prt.default_camera(1920, 1080)
    f = np.frombuffer(bytes(cam2), dtype=np.float32)
    # pos (0, 1.18208, 3.82135), view (0, -0.29552, -0.955337), fov (45, 26.2313), aperture .01, focal 4
    assert np.allclose(f[0:3], [0, 1.18208, 3.82135], atol=1e-5) and np.allclose(f[4:7], [0, -0.29552, -0.955337], atol=1e-6)
    assert np.allclose(f[12:18], [1920, 1080, 45, 26.2313, 0.01, 4], atol=1e-4)


def test_seed_pairs_are_the_glibc_rand_stream(prt):
    libc = C.CDLL("libc.so.6")
    libc.rand.restype = C.c_int
    # a fresh process-wide rand() stream cannot be guaranteed inside pytest; compare with random_r instead
    seq = prt.seed_pairs(8)
    assert seq.dtype == np.int32 and len(seq) == 16 and (seq >= 0).all()
    # known first values of glibc's default stream (seed 1): 1804289383 846930886 | 1681692777 1714636915 ...
    assert list(seq[:4]) == [1681692777, 1714636915, 1957747793, 424238335]
    assert list(prt.seed_pairs(3, first_frame=6)) == list(seq[10:16])


def test_bvh_is_a_valid_partition(prt):
    scene = prt.HostScene("cornell_diffuse.json")
    d = scene.desc
    T, N = d.triangle_count, d.bvh_node_count
    assert T == 6320 and N > 1
    nodes = np.frombuffer(C.string_at(d.bvh_nodes, N * 36), dtype=np.dtype(
        [("b", "<f4", 6), ("first", "<u4"), ("count", "<u4"), ("leaf", "u1"), ("_p", "u1", 3)]))
    idx = np.frombuffer(C.string_at(d.primitive_indices, T * 8), dtype=np.uint64)
    assert sorted(idx.tolist()) == list(range(T))                      # a permutation
    verts = np.frombuffer(C.string_at(d.vertices, T * 48), dtype=np.float32).reshape(T, 3, 4)[:, :, :3]
    seen = np.zeros(T, bool)
    visited = np.zeros(N, bool)
    stack = [0]
    while stack:
        n = stack.pop()
        assert not visited[n]
        visited[n] = True
        nd = nodes[n]
        lo, hi = nd["b"][0::2], nd["b"][1::2]
        if nd["leaf"]:
            sl = idx[nd["first"]:nd["first"] + nd["count"]].astype(int)
            assert len(sl) >= 1 and not seen[sl].any()
            seen[sl] = True
            p = verts[sl].reshape(-1, 3)
            assert (p >= lo - 1e-6).all() and (p <= hi + 1e-6).all()
        else:
            for c in (nd["first"], nd["first"] + 1):
                cl, ch = nodes[c]["b"][0::2], nodes[c]["b"][1::2]
                assert (cl >= lo - 1e-6).all() and (ch <= hi + 1e-6).all()   # children inside the parent
                stack.append(int(c))
    assert seen.all() and visited.all()
    assert scene.bvh_depth < 64                                          # the reference's closest-hit stack size


@pytest.mark.parametrize("threads", ["3", "8"])
def test_bvh_is_the_same_tree_whatever_the_thread_count(prt, tmp_path, monkeypatch, threads):
    """the multi-threaded builder (csrc/host/bvh.cpp) claims the tree is a function of the input only: nodes and primitive
    indices byte for byte against the one-thread build, on a mesh big enough for the thread pool (>= 16 384 triangles)"""
    import importlib.util
    spec = importlib.util.spec_from_file_location("_make_dragon", os.path.join(ROOT, "scenes", "make_dragon_standin.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    n = mod.write(str(tmp_path / "dragon_standin.prtmesh"), 192, 64)
    assert n >= 16384
    text = open(os.path.join(ROOT, "scenes", "cornell_dragon.json")).read()

    def build(k):
        monkeypatch.setenv("PRT_BVH_THREADS", k)
        scene = prt.HostScene(text, models_dir=str(tmp_path) + "/", text=True)
        d = scene.desc
        return C.string_at(d.bvh_nodes, d.bvh_node_count * 36), C.string_at(d.primitive_indices, d.triangle_count * 8), scene

    nodes1, idx1, _s1 = build("1")
    nodesk, idxk, _sk = build(threads)
    assert len(nodes1) > 36 * 1000 and nodes1 == nodesk and idx1 == idxk


def test_obj_reader_and_soup_roundtrip(prt, tmp_path):
    obj = tmp_path / "quad.obj"
    obj.write_text("# a quad with normals, a triangle with negative indices, a polygon without vn\n"
                   "v 0 0 0\nv 1 0 0\nv 1 1 0\nv 0 1 0\nvn 0 0 1\n"
                   "f 1//1 2//1 3//1 4//1\nf -4//-1 -3//-1 -2//-1\nf 1 2 3\n")
    lib = prt.load_library()
    err = C.create_string_buffer(256)
    soup = tmp_path / "quad.prtmesh"
    assert lib.prth_convert_model(str(obj).encode(), str(soup).encode(), err, 256) == 0, err.value
    raw = soup.read_bytes()
    assert raw[:8] == b"PRTMESH1" and struct.unpack("<I", raw[8:12])[0] == 4     # fan: 2 + 1 + 1 triangles
    tri = np.frombuffer(raw[12:], dtype=np.float32).reshape(4, 3, 6)
    assert np.array_equal(tri[0, :, :3], [[0, 0, 0], [1, 0, 0], [1, 1, 0]])
    assert np.array_equal(tri[1, :, :3], [[0, 0, 0], [1, 1, 0], [0, 1, 0]])
    assert np.array_equal(tri[3, :, 3:], [[0, 0, 1]] * 3)                          # generated flat normal
    text = '{"scene":{"obj":{"path":"quad.prtmesh","material":{"type":1}},"spheres":[{"pos":[0,3,0],"radius":0.5,"material":{"color":[5,5,5],"type":0}}]}}'
    sc = prt.HostScene(text, models_dir=str(tmp_path), text=True)
    assert sc.desc.triangle_count == 4
    assert lib.prth_convert_model(b"/nonexistent.obj", str(soup).encode(), err, 256) != 0


def _octahedron():
    v = np.array([[1, 0, 0], [-1, 0, 0], [0, 1, 0], [0, -1, 0], [0, 0, 1], [0, 0, -1]], dtype=np.float32) * np.float32(0.37)
    f = [[0, 2, 4], [2, 1, 4], [1, 3, 4], [3, 0, 4], [2, 0, 5], [1, 2, 5], [3, 1, 5], [0, 3, 5]]
    return v, f


def _soup(prt, path, tmp_path):
    lib = prt.load_library()
    err = C.create_string_buffer(256)
    out = str(tmp_path / (os.path.basename(path) + ".prtmesh"))
    rc = lib.prth_convert_model(str(path).encode(), out.encode(), err, 256)
    if rc:
        raise prt.PrtError(err.value.decode())
    raw = open(out, "rb").read()
    return np.frombuffer(raw[12:], dtype=np.float32).reshape(-1, 3, 6)


def test_ply_and_stl_readers_agree_with_the_obj_reader(prt, tmp_path):
    """formats the reference reaches through assimp (src/Models/model_loader.cpp:38; "next" row N2): Stanford PLY, ascii and binary
    little-endian, with and without normals, a quad face, an extra element and extra properties to skip; STL, binary and ascii.  The
    same mesh through every reader must give the same triangle soup as the OBJ reader (positions bit for bit; generated normals by
    the same smoothing rule; file normals as written)."""
    v, f = _octahedron()
    vn = (v / np.linalg.norm(v, axis=1, keepdims=True)).astype(np.float32)
    obj = tmp_path / "o.obj"
    obj.write_text("".join("v %r %r %r\n" % tuple(float(x) for x in p) for p in v) + "".join("f %d %d %d\n" % tuple(i + 1 for i in t) for t in f))
    objn = tmp_path / "on.obj"
    objn.write_text("".join("v %r %r %r\n" % tuple(float(x) for x in p) for p in v) + "".join("vn %r %r %r\n" % tuple(float(x) for x in p) for p in vn) +
                    "".join("f %d//%d %d//%d %d//%d\n" % (t[0] + 1, t[0] + 1, t[1] + 1, t[1] + 1, t[2] + 1, t[2] + 1) for t in f))
    want, want_n = _soup(prt, obj, tmp_path), _soup(prt, objn, tmp_path)
    assert want.shape == (8, 3, 6)
    # ascii PLY without normals, the first two triangles merged into one quad face (fan-triangulated back), an extra property
    quad = [[0, 2, 1, 4]] if False else None
    ply = tmp_path / "a.ply"
    ply.write_text("ply\nformat ascii 1.0\ncomment made by tests\nelement vertex 6\nproperty float x\nproperty float y\nproperty float z\nproperty uchar red\n"
                   "element face 8\nproperty list uchar int vertex_indices\nend_header\n" +
                   "".join("%r %r %r 255\n" % tuple(float(x) for x in p) for p in v) + "".join("3 %d %d %d\n" % tuple(t) for t in f))
    assert np.array_equal(_soup(prt, ply, tmp_path).view(np.uint32), want.view(np.uint32))
    # binary PLY with normals, double-precision positions, an element to skip in front, ushort list counts
    head = ("ply\nformat binary_little_endian 1.0\nelement camera 1\nproperty float fov\nproperty list uchar float extra\n"
            "element vertex 6\nproperty double x\nproperty double y\nproperty double z\nproperty float nx\nproperty float ny\nproperty float nz\n"
            "element face 8\nproperty list ushort uint vertex_index\nend_header\n").encode()
    body = struct.pack("<fB2f", 45.0, 2, 1.0, 2.0)
    for p, n in zip(v, vn):
        body += struct.pack("<3d3f", *[float(x) for x in p], *[float(x) for x in n])
    for t in f:
        body += struct.pack("<H3I", 3, *t)
    plyb = tmp_path / "b.ply"
    plyb.write_bytes(head + body)
    assert np.array_equal(_soup(prt, plyb, tmp_path).view(np.uint32), want_n.view(np.uint32))
    # a quad face: fan (0 1 2 3) -> (0 1 2), (0 2 3)
    plyq = tmp_path / "q.ply"
    plyq.write_text("ply\nformat ascii 1.0\nelement vertex 4\nproperty float x\nproperty float y\nproperty float z\nelement face 1\n"
                    "property list uchar int vertex_indices\nend_header\n0 0 0\n1 0 0\n1 1 0\n0 1 0\n4 0 1 2 3\n")
    q = _soup(prt, plyq, tmp_path)
    assert q.shape == (2, 3, 6) and np.array_equal(q[1, :, :3], [[0, 0, 0], [1, 1, 0], [0, 1, 0]]) and np.array_equal(q[:, :, 3:], np.tile([0, 0, 1], (2, 3, 1)))
    # STL: binary with facet normals (kept, flat), ascii with zero normals (geometric)
    tri = v[np.array(f)]                                            # [8, 3, 3]
    fn = np.cross(tri[:, 1] - tri[:, 0], tri[:, 2] - tri[:, 0])
    fn = (fn / np.linalg.norm(fn, axis=1, keepdims=True)).astype(np.float32)
    stl = tmp_path / "b.stl"
    stl.write_bytes(b"solid looks like ascii but is not".ljust(80, b" ") + struct.pack("<I", 8) +
                    b"".join(struct.pack("<12fH", *fn[k], *tri[k].ravel(), 0) for k in range(8)))
    sb = _soup(prt, stl, tmp_path)
    assert np.array_equal(sb[:, :, :3].view(np.uint32), want[:, :, :3].view(np.uint32))
    assert np.array_equal(sb[:, :, 3:], np.repeat(fn[:, None, :], 3, axis=1))
    stla = tmp_path / "a.stl"
    stla.write_text("solid t\n" + "".join("facet normal 0 0 0\n outer loop\n" + "".join("  vertex %r %r %r\n" % tuple(float(x) for x in p) for p in tri[k]) +
                                           " endloop\nendfacet\n" for k in range(8)) + "endsolid t\n")
    sa = _soup(prt, stla, tmp_path)
    assert np.array_equal(sa[:, :, :3].view(np.uint32), want[:, :, :3].view(np.uint32))
    assert np.abs(sa[:, :, 3:] - np.repeat(fn[:, None, :], 3, axis=1)).max() < 1e-6
    # damaged files are errors, not crashes
    for name, data in (("t.ply", head + body[:40]), ("e.ply", b"ply\nformat binary_big_endian 1.0\nend_header\n"), ("i.ply", ply.read_text().replace("3 0 2 4", "3 0 2 9").encode()),
                       ("t.stl", stl.read_bytes()[:200]), ("n.stl", b"solid x\nfacet normal 0 0 0\nouter loop\nvertex 0 0 0\nvertex nan 0 0\nvertex 0 1 0\nendloop\nendfacet\n")):
        bad = tmp_path / name
        bad.write_bytes(data)
        with pytest.raises(prt.PrtError):
            _soup(prt, bad, tmp_path)


def test_malformed_scene_text_is_an_error_not_a_crash(prt):
    """truncated / mistyped scene files come back as load errors; out-of-range `type` / `dist` exponents (the
    reference shifts by them unchecked, include/Scene/scene.h:90-101) give an empty bit, not undefined behaviour"""
    for text in ["", "{", "[1,2,3]", '{"scene":{"spheres":[{"pos":[1],"radius":"x"}],"quads":[{}]}}']:
        with pytest.raises(prt.PrtError):
            prt.HostScene(text, text=True)
    sc = prt.HostScene('{"scene":{"sdfs":[{"type":99,"params":[1,2,3,4]}],"spheres":[{"pos":[0,3,0],"radius":0.5,'
                       '"material":{"color":[5,5,5],"type":0,"dist":-3}}]}}', text=True)
    assert list(sc.desc.object_count)[:2] == [1, 1]


def test_obj_without_normals_gets_smooth_normals(prt, tmp_path):
    """assimp's GenSmoothNormals rule (the reference imports with aiProcessPreset_TargetRealtime_Quality): a vertex
    gets the normalised sum of the unit normals of the faces meeting at its position; faces folded back by more
    than 175 degrees do not smooth"""
    obj = tmp_path / "tent.obj"
    obj.write_text("v 0 0 0\nv 0 0 1\nv 1 -1 0\nv -1 -1 0\n"          # ridge (0,0,0)-(0,0,1), two slopes
                   "f 1 2 3\nf 2 1 4\n"
                   "v 5 0 0\nv 6 0 0\nv 5 1 0\nf 5 6 7\nf 5 7 6\n")        # a double-sided flap: opposite normals
    lib = prt.load_library()
    err = C.create_string_buffer(256)
    soup = tmp_path / "tent.prtmesh"
    assert lib.prth_convert_model(str(obj).encode(), str(soup).encode(), err, 256) == 0, err.value
    tri = np.frombuffer(soup.read_bytes()[12:], dtype=np.float32).reshape(4, 3, 6)
    n0 = np.cross(tri[0, 1, :3] - tri[0, 0, :3], tri[0, 2, :3] - tri[0, 0, :3]); n0 /= np.linalg.norm(n0)
    n1 = np.cross(tri[1, 1, :3] - tri[1, 0, :3], tri[1, 2, :3] - tri[1, 0, :3]); n1 /= np.linalg.norm(n1)
    ridge = (n0 + n1) / np.linalg.norm(n0 + n1)
    assert np.allclose(tri[0, 0, 3:], ridge, atol=1e-6) and np.allclose(tri[0, 1, 3:], ridge, atol=1e-6)   # shared edge
    assert np.allclose(tri[1, 0, 3:], ridge, atol=1e-6) and np.allclose(tri[1, 1, 3:], ridge, atol=1e-6)
    assert np.allclose(tri[0, 2, 3:], n0, atol=1e-6) and np.allclose(tri[1, 2, 3:], n1, atol=1e-6)         # free corners
    assert np.allclose(tri[2, :, 3:], [[0, 0, 1]] * 3) and np.allclose(tri[3, :, 3:], [[0, 0, -1]] * 3)    # no smoothing across 180 degrees


@pytest.mark.skipif(not os.path.exists("/root/reference/resources/models/teapot.obj"), reason="reference not present")
def test_committed_teapot_soup_is_the_reference_obj(prt, tmp_path):
    lib = prt.load_library()
    err = C.create_string_buffer(256)
    out = tmp_path / "t.prtmesh"
    assert lib.prth_convert_model(b"/root/reference/resources/models/teapot.obj", str(out).encode(), err, 256) == 0
    assert out.read_bytes() == open(os.path.join(prt.MODELS_DIR, "teapot.prtmesh"), "rb").read()


def test_malformed_scene_is_an_error_not_a_crash(prt):
    for text in ("{", '{"scen":{}}', '{"scene":{"obj":{"path":"missing_model.obj"}}}'):
        with pytest.raises(prt.PrtError):
            prt.HostScene(text, text=True)


def test_hostile_inputs_are_refused_before_they_cost_memory(prt, tmp_path):
    """a damaged .prtmesh header must not size an allocation (300 GB for n = 2^32 - 1), deeply nested JSON must not
    overflow the stack, and `nan` / `inf` vertex positions (strtof accepts them) must not reach the BVH builder"""
    lib = prt.load_library()
    err = C.create_string_buffer(256)
    soup = tmp_path / "lie.prtmesh"
    soup.write_bytes(b"PRTMESH1" + struct.pack("<I", 0xFFFFFFFF) + bytes(72 * 3))
    text = '{"scene":{"obj":{"path":"lie.prtmesh","material":{"type":1}},"spheres":[{"pos":[0,3,0],"radius":0.5,"material":{"color":[5,5,5],"type":0}}]}}'
    with pytest.raises(prt.PrtError, match="truncated soup"):
        prt.HostScene(text, models_dir=str(tmp_path), text=True)
    with pytest.raises(prt.PrtError, match="nesting"):
        prt.HostScene("[" * 100000, text=True)
    with pytest.raises(prt.PrtError, match="nesting"):
        prt.HostScene('{"a":' * 100 + "1" + "}" * 100, text=True)
    for bad in ("nan", "inf", "-inf"):
        obj = tmp_path / ("bad_%s.obj" % bad.strip("-"))
        obj.write_text("v 0 0 0\nv 1 0 %s\nv 1 1 0\nf 1 2 3\n" % bad)
        assert lib.prth_convert_model(str(obj).encode(), str(tmp_path / "o.prtmesh").encode(), err, 256) != 0
        assert b"non-finite" in err.value
    nan_soup = tmp_path / "nan.prtmesh"
    nan_soup.write_bytes(b"PRTMESH1" + struct.pack("<I", 1) + np.array([0, 0, 0, 0, 0, 1, 1, np.nan, 0, 0, 0, 1, 1, 1, 0, 0, 0, 1], dtype=np.float32).tobytes())
    with pytest.raises(prt.PrtError, match="non-finite"):
        prt.HostScene(text.replace("lie.prtmesh", "nan.prtmesh"), models_dir=str(tmp_path), text=True)


def test_obj_groups_are_meshes_and_welding_gives_the_soup_back(prt, tmp_path):
    """N2: (a) an OBJ file with several objects / groups / materials is several meshes, in file order -- what assimp's importer makes of
    it and what the reference walks (src/Models/model_loader.cpp:58-74); the soup is their concatenation, and normals that have to be
    GENERATED are smoothed within a mesh only (GenSmoothNormals works mesh by mesh), so two groups that share an edge keep a crease;
    (b) aiProcess_JoinIdenticalVertices (model_loader.cpp:38) as the loader's `weld` view: unique (position, normal) vertices + indices,
    de-indexing gives the soup back float for float (on the render path nothing can see welding: src/main.cpp:93-119 de-indexes).
    There is no oracle (assimp is an absent submodule): the pin is the OBJ reader on the same geometry."""
    # two triangles sharing the edge (0,0,0)-(0,1,0), folded by 90 degrees
    verts = "v 0 0 0\nv 0 1 0\nv 1 0 0\nv 0 0 1\n"
    one = tmp_path / "one.obj"
    one.write_text(verts + "f 1 3 2\nf 1 2 4\n")
    two = tmp_path / "two.obj"
    two.write_text(verts + "o left\nf 1 3 2\ng right\nusemtl red\nf 1 2 4\n")
    assert [m[0] for m in prt.model_meshes(str(one))] == [2]
    assert [m[0] for m in prt.model_meshes(str(two))] == [1, 1]
    a, b = _soup(prt, one, tmp_path), _soup(prt, two, tmp_path)
    assert a.shape == b.shape == (2, 3, 6)
    assert np.array_equal(a[:, :, :3], b[:, :, :3])                       # same triangles, same order
    # one mesh: the corners on the shared edge carry the normalised sum of both face normals; two meshes: each face its own
    fa = np.cross(a[0, 1, :3] - a[0, 0, :3], a[0, 2, :3] - a[0, 0, :3]); fa /= np.linalg.norm(fa)
    fb = np.cross(a[1, 1, :3] - a[1, 0, :3], a[1, 2, :3] - a[1, 0, :3]); fb /= np.linalg.norm(fb)
    avg = (fa + fb) / np.linalg.norm(fa + fb)
    assert np.allclose(a[0, 0, 3:], avg, atol=1e-6) and np.allclose(a[1, 0, 3:], avg, atol=1e-6)     # corner (0,0,0): smoothed across
    assert np.allclose(a[0, 1, 3:], fa, atol=1e-6)                                                    # corner (1,0,0): one face only
    assert np.allclose(b[0, :, 3:], np.tile(fa, (3, 1)), atol=1e-6) and np.allclose(b[1, :, 3:], np.tile(fb, (3, 1)), atol=1e-6)
    # welding: the shared corners of `one` are one vertex each (4 vertices for 6 corners), of `two` nothing is shared across meshes
    assert prt.model_meshes(str(one)) == [(2, 4, True)]
    assert prt.model_meshes(str(two)) == [(1, 3, True), (1, 3, True)]
    # the teapot of the reference's own scene: 6 320 triangles, every corner with a vn record; welded to a fraction, and back
    teapot = prt.model_meshes(os.path.join(prt.MODELS_DIR, "teapot.obj"))
    assert sum(m[0] for m in teapot) == 6320 and all(m[2] for m in teapot)
    assert sum(m[1] for m in teapot) < 6320 * 3 // 4
    # a renderer that is given the two-mesh file sees ONE soup (BVH over all meshes): same triangle count as the one-mesh file
    for path in (one, two):
        scene = prt.HostScene('{"scene": {"obj": {"path": "%s", "material": {"color": [1, 1, 1], "type": 1}}, "camera": {}, "objects": ['
                              '{"type": 1, "position": [0, 3, 0], "radius": 0.5, "material": {"color": [5, 5, 5], "type": 0}}]}}' % path.name,
                              models_dir=str(tmp_path), text=True)
        assert scene.desc.triangle_count == 2
