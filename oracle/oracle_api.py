"""oracle/oracle_api.py -- TEST INFRASTRUCTURE.  ctypes access to the two CPU checkers:

  * RefOracle        oracle/_ref/libref_<variant>.so  -- the reference's own kernel text compiled
                     for the host (built by oracle/ref/build_ref.py from /root/reference, only in
                     the development container)
  * Restatement      oracle/liboracle_pt.so           -- oracle/pt_oracle.c, the plain-C restatement
                     of the reference algorithm (travels everywhere; built by oracle/Makefile)

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
The product (libprt.so and the package next to it) never does.
"""
import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF_DIR = os.path.join(HERE, "_ref")

PATH_STATE_DTYPE = np.dtype([("origin", "<f4", 4), ("dir", "<f4", 4), ("time", "<f4"), ("dist", "<f4"), ("_p0", "u1", 8),
                             ("mask", "<f4", 4), ("acc", "<f4", 4), ("total", "<u4"),
                             ("diff", "<u2"), ("spec", "<u2"), ("trans", "<u2"), ("scatters", "<u2"),
                             ("was_specular", "u1"), ("_p1", "u1", 3), ("reset", "u1"), ("_p2", "u1", 3),
                             ("samples", "<u4"), ("_p3", "u1", 8)])
assert PATH_STATE_DTYPE.itemsize == 112

STATE_FIELDS = ["origin", "dir", "time", "dist", "mask", "acc", "total", "diff", "spec", "trans", "scatters",
                "was_specular", "reset", "samples"]


def float_bits(x):
    """bit patterns for exact comparison: -0 != +0, every NaN compares equal to every NaN (NaN
    payload/sign propagation is the one thing IEEE 754 leaves to the implementation; it only
    occurs in degenerate set-ups such as a 1-pixel-wide image, camera.cl:32 divides by W-1)"""
    x = np.ascontiguousarray(x, dtype=np.float32)
    b = x.view(np.uint32).copy()
    b[np.isnan(x)] = 0x7FC00000
    return b


def images_equal(a, b):
    return np.array_equal(float_bits(a), float_bits(b))


def state_fields_equal(a, b):
    """bit-exact comparison of two RTD arrays, field by field (padding bytes and the unused w lanes
    of the float3 members are not state).  Returns the list of differing field names."""
    bad = []
    for f in STATE_FIELDS:
        x, y = a[f], b[f]
        if f in ("origin", "dir", "mask"):
            x, y = x[..., :3], y[..., :3]
        if x.dtype.kind == "f":
            x, y = float_bits(x), float_bits(y)
        if not np.array_equal(x, y):
            bad.append(f)
    return bad


def ref_available(variant):
    return os.path.exists(os.path.join(REF_DIR, "libref_%s.so" % variant))


class _SceneArrays:
    """numpy copies of the buffers of a prt_scene_desc (so the oracle runs on exactly what the
    product is given)."""

    def __init__(self, desc):
        n = desc.object_count[7]
        self.n_meshes = n
        self.meshes = np.frombuffer(C.string_at(desc.meshes, n * 256), dtype=np.uint8).copy()
        self.counts = np.array(list(desc.object_count), dtype=np.uint32)
        self.obj_mat = np.frombuffer(C.string_at(desc.obj_material, 64), dtype=np.uint8).copy()
        t = desc.triangle_count
        self.n_tris = t
        self.vertices = np.frombuffer(C.string_at(desc.vertices, t * 48), dtype=np.float32).copy() if t else np.zeros(4, np.float32)
        self.normals = np.frombuffer(C.string_at(desc.normals, t * 48), dtype=np.float32).copy() if t else np.zeros(4, np.float32)
        self.indices = np.frombuffer(C.string_at(desc.primitive_indices, t * 8), dtype=np.uint64).copy() if t else np.zeros(1, np.uint64)
        nn = desc.bvh_node_count
        self.n_nodes = nn
        self.nodes = np.frombuffer(C.string_at(desc.bvh_nodes, nn * 36), dtype=np.uint8).copy()


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


class RefOracle:
    """One scene-specialised build of the reference kernel (the scene decides the variant)."""

    def __init__(self, variant):
        path = os.path.join(REF_DIR, "libref_%s.so" % variant)
        self.lib = C.CDLL(path)
        self.lib.ref_render_frames.restype = C.c_int
        self.lib.ref_render_frames.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_uint, C.c_int,
                                               C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                               C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int,
                                               C.c_void_p, C.c_void_p, C.c_uint, C.c_int]

    def render(self, desc, camera_bytes, W, H, seed_pairs, first_frame=1, state=None, env=None,
               spp_limit=0, threads=8):
        sa = desc if isinstance(desc, _SceneArrays) else _SceneArrays(desc)
        n_frames = len(seed_pairs) // 2
        if state is None:
            state = np.zeros(W * H, dtype=PATH_STATE_DTYPE)
        img = np.zeros((H, W, 4), dtype=np.float32)
        cam = np.frombuffer(camera_bytes, dtype=np.uint8).copy()
        seeds = np.asarray(seed_pairs, dtype=np.int32)
        if env is not None:
            env = np.ascontiguousarray(env, dtype=np.float32)
            eh, ew = env.shape[0], env.shape[1]
            envp = _ptr(env)
        else:
            eh = ew = 0
            envp = None
        rc = self.lib.ref_render_frames(_ptr(sa.meshes), sa.n_meshes, _ptr(sa.counts), W, H, first_frame, n_frames,
                                        _ptr(seeds), _ptr(cam), _ptr(sa.indices), _ptr(sa.vertices), _ptr(sa.normals),
                                        _ptr(sa.obj_mat), _ptr(sa.nodes), envp, ew, eh, _ptr(state), _ptr(img),
                                        spp_limit, threads)
        assert rc == 0
        return state, img


SceneArrays = _SceneArrays


class _Job(C.Structure):
    _fields_ = [("cfg", C.c_void_p), ("scene", C.c_void_p), ("camera", C.c_void_p),
                ("env_rgb", C.c_void_p), ("env_w", C.c_int), ("env_h", C.c_int),
                ("width", C.c_int), ("full_height", C.c_int), ("row0", C.c_int), ("rows", C.c_int),
                ("block_rows", C.c_int), ("n_parts", C.c_int), ("part", C.c_int),
                ("first_frame", C.c_uint32), ("n_frames", C.c_uint32), ("seed_pairs", C.c_void_p),
                ("state", C.c_void_p), ("out_rgba", C.c_void_p), ("spp_limit", C.c_uint32), ("n_threads", C.c_int)]


class _Diag(C.Structure):
    _fields_ = [("max_stack", C.c_int), ("max_shadow_stack", C.c_int)]


class Restatement:
    """oracle/liboracle_pt.so (pt_oracle.c).  `cfg`, `desc`, `camera` are the ctypes structs of
    include/prt.h (prt_config, prt_scene_desc, prt_camera) -- the same objects the product gets."""

    def __init__(self, path=None):
        path = path or os.path.join(HERE, "liboracle_pt.so")
        if not os.path.exists(path):
            raise RuntimeError("oracle/liboracle_pt.so not built (make -C oracle)")
        self.lib = C.CDLL(path)
        self.lib.pto_render.restype = C.c_int
        self.lib.pto_render.argtypes = [C.POINTER(_Job), C.POINTER(_Diag)]
        self.last_diag = None

    def render(self, cfg, desc, camera, W, H, seed_pairs, first_frame=1, state=None, env=None,
               spp_limit=0, threads=8, row0=0, rows=None, blocks=None):
        """blocks = (block_rows, n_parts, part) selects interleaved row blocks (prt_set_row_blocks)"""
        if blocks is not None:
            rows = sum(1 for r in range(H) if (r // blocks[0]) % blocks[1] == blocks[2])
        rows = H if rows is None else rows
        n_frames = len(seed_pairs) // 2
        if state is None:
            state = np.zeros(W * rows, dtype=PATH_STATE_DTYPE)
        img = np.zeros((rows, W, 4), dtype=np.float32)
        seeds = np.ascontiguousarray(seed_pairs, dtype=np.int32)
        job = _Job()
        job.cfg = C.cast(C.pointer(cfg), C.c_void_p)
        job.scene = C.cast(C.pointer(desc), C.c_void_p)
        job.camera = C.cast(C.pointer(camera), C.c_void_p)
        if env is not None:
            env = np.ascontiguousarray(env, dtype=np.float32)
            job.env_rgb = _ptr(env)
            job.env_h, job.env_w = env.shape[0], env.shape[1]
        job.width, job.full_height, job.row0, job.rows = W, H, row0, rows
        job.block_rows, job.n_parts, job.part = blocks if blocks is not None else (1, 1, 0)
        job.first_frame, job.n_frames = first_frame, n_frames
        job.seed_pairs = _ptr(seeds)
        job.state = _ptr(state)
        job.out_rgba = _ptr(img)
        job.spp_limit = spp_limit
        job.n_threads = threads
        diag = _Diag()
        rc = self.lib.pto_render(C.byref(job), C.byref(diag))
        if rc != 0:
            raise RuntimeError("pto_render failed: %d" % rc)
        self.last_diag = (diag.max_stack, diag.max_shadow_stack)
        return state, img
