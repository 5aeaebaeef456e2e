"""ctypes view of include/prt.h, include/prt_types.h and csrc/host/host_capi.h.

The product is the C-ABI shared library `libprt.so` (HIP kernels + host model).  This module
only declares the structs and prototypes so Python plumbing (tests, bench.py, torch.distributed
launch) can call it; nothing here computes pixels and there is no Python or CPU fallback.
"""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("PRT_LIB", os.path.join(HERE, "libprt.so"))   # PRT_LIB: A/B builds of the same product

PRT_MAX_LIGHTS = 16
PRT_ABI_VERSION = 3


class Material(C.Structure):
    _fields_ = [("color", C.c_float * 4), ("eta", C.c_float * 4), ("k", C.c_float * 4),
                ("roughness", C.c_float), ("t", C.c_uint16), ("lobes", C.c_uint8), ("dist", C.c_uint8),
                ("_pad", C.c_uint8 * 8)]


class Mesh(C.Structure):
    _fields_ = [("mat", Material), ("pos", C.c_float * 4), ("_pad0", C.c_uint8 * 48),
                ("joker", C.c_float * 16), ("t", C.c_uint8), ("_pad1", C.c_uint8 * 63)]


class BvhNode(C.Structure):
    _fields_ = [("bounds", C.c_float * 6), ("first_child_or_primitive", C.c_uint32),
                ("primitive_count", C.c_uint32), ("is_leaf", C.c_uint8), ("_pad", C.c_uint8 * 3)]


class Camera(C.Structure):
    _fields_ = [("position", C.c_float * 4), ("view", C.c_float * 4), ("up", C.c_float * 4),
                ("resolution", C.c_float * 2), ("fov", C.c_float * 2), ("apertureRadius", C.c_float),
                ("focalDistance", C.c_float), ("_pad", C.c_uint8 * 8)]


class Config(C.Structure):
    _fields_ = [("abi_version", C.c_uint32),
                ("max_bounces", C.c_int32), ("max_diff_bounces", C.c_int32), ("max_spec_bounces", C.c_int32),
                ("max_trans_bounces", C.c_int32), ("max_scattering_events", C.c_int32),
                ("marching_steps", C.c_int32), ("shadow_marching_steps", C.c_int32),
                ("active_mats", C.c_uint32), ("geom_flags", C.c_uint32), ("light_count", C.c_uint32),
                ("light_indices", C.c_uint32 * PRT_MAX_LIGHTS),
                ("has_global_medium", C.c_int32), ("fog_density", C.c_float), ("fog_sigma_a", C.c_float),
                ("fog_sigma_s", C.c_float), ("fog_sigma_t", C.c_float), ("fog_abs_only", C.c_int32),
                ("alpha_testing", C.c_int32), ("phase_function", C.c_int32), ("phase_g", C.c_float),
                ("view_option", C.c_uint32), ("pick_random_light", C.c_uint32), ("env_importance_sampling", C.c_uint32)]


class SceneDesc(C.Structure):
    _fields_ = [("meshes", C.c_void_p), ("object_count", C.c_uint32 * 8), ("obj_material", C.c_void_p),
                ("vertices", C.c_void_p), ("normals", C.c_void_p), ("primitive_indices", C.c_void_p),
                ("triangle_count", C.c_uint32), ("bvh_nodes", C.c_void_p), ("bvh_node_count", C.c_uint32)]


class Stats(C.Structure):
    _fields_ = [("kernel_ms", C.c_double), ("launches", C.c_uint32), ("frames", C.c_uint32),
                ("samples", C.c_uint64), ("segments", C.c_uint64), ("finished_pixels", C.c_uint64),
                ("kernel_sum_ms", C.c_double), ("concurrent", C.c_uint32), ("_pad", C.c_uint32)]


assert C.sizeof(Material) == 64 and C.sizeof(Mesh) == 256 and C.sizeof(BvhNode) == 36 and C.sizeof(Camera) == 80

# numpy view of the 112-byte RTD (prt_path_state)
PATH_STATE_DTYPE = [("origin", "<f4", 4), ("dir", "<f4", 4), ("time", "<f4"), ("dist", "<f4"), ("_p0", "u1", 8),
                    ("mask", "<f4", 4), ("acc", "<f4", 4), ("total", "<u4"),
                    ("diff", "<u2"), ("spec", "<u2"), ("trans", "<u2"), ("scatters", "<u2"),
                    ("was_specular", "u1"), ("_p1", "u1", 3), ("reset", "u1"), ("_p2", "u1", 3),
                    ("samples", "<u4"), ("_p3", "u1", 8)]

# (name, restype, argtypes) of every symbol include/prt.h declares
PRT_API = [
    ("prt_create", C.c_int, [C.c_int, C.POINTER(Config), C.POINTER(C.c_void_p)]),
    ("prt_destroy", None, [C.c_void_p]),
    ("prt_upload_scene", C.c_int, [C.c_void_p, C.POINTER(SceneDesc)]),
    ("prt_set_camera", C.c_int, [C.c_void_p, C.POINTER(Camera)]),
    ("prt_upload_envmap", C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int]),
    ("prt_resize", C.c_int, [C.c_void_p, C.c_int, C.c_int]),
    ("prt_set_tile", C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int]),
    ("prt_set_row_blocks", C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]),
    ("prt_reset", C.c_int, [C.c_void_p]),
    ("prt_render_frames", C.c_int, [C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p]),
    ("prt_render_spp", C.c_int, [C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.POINTER(C.c_uint32)]),
    ("prt_set_walk_min_lanes", C.c_int, [C.c_void_p, C.c_uint32]),
    ("prt_set_option", C.c_int, [C.c_void_p, C.c_char_p, C.c_int]),
    ("prt_kernel_variant", C.c_char_p, [C.c_void_p]),
    ("prt_build_id", C.c_char_p, []),
    ("prt_synchronize", C.c_int, [C.c_void_p]),
    ("prt_read_framebuffer", C.c_int, [C.c_void_p, C.c_void_p]),
    ("prt_tonemap_rgba8", C.c_int, [C.c_void_p, C.c_void_p]),
    ("prt_copy_framebuffer_to_device", C.c_int, [C.c_void_p, C.c_void_p]),
    ("prt_read_state", C.c_int, [C.c_void_p, C.c_void_p]),
    ("prt_write_state", C.c_int, [C.c_void_p, C.c_void_p]),
    ("prt_set_stream", C.c_int, [C.c_void_p, C.c_void_p]),
    ("prt_get_stats", C.c_int, [C.c_void_p, C.POINTER(Stats)]),
    ("prt_query_counts", C.c_int, [C.c_void_p, C.c_uint32, C.POINTER(Stats)]),
    ("prt_selftest_math", C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]),
    ("prt_selftest_fn", C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]),
    ("prt_last_error", C.c_char_p, [C.c_void_p]),
    ("prt_last_global_error", C.c_char_p, []),
]

PRTH_API = [
    ("prth_scene_load", C.c_void_p, [C.c_char_p, C.c_char_p, C.c_char_p, C.c_int]),
    ("prth_scene_load_text", C.c_void_p, [C.c_char_p, C.c_char_p, C.c_char_p, C.c_int]),
    ("prth_scene_free", None, [C.c_void_p]),
    ("prth_scene_get_desc", C.c_int, [C.c_void_p, C.POINTER(SceneDesc)]),
    ("prth_scene_get_config", C.c_int, [C.c_void_p, C.c_int, C.POINTER(Config)]),
    ("prth_scene_bvh_depth", C.c_int, [C.c_void_p]),
    ("prth_scene_obj_path", C.c_char_p, [C.c_void_p]),
    ("prth_default_camera", C.c_int, [C.c_int, C.c_int, C.c_float, C.POINTER(Camera)]),
    ("prth_orbit_camera", C.c_int, [C.c_int, C.c_int, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float,
                                    C.c_float, C.POINTER(Camera)]),
    ("prth_seed_pairs", C.c_int, [C.c_uint32, C.c_uint32, C.c_void_p]),
    ("prth_convert_model", C.c_int, [C.c_char_p, C.c_char_p, C.c_char_p, C.c_int]),
    ("prth_model_meshes", C.c_int, [C.c_char_p, C.c_void_p, C.c_int, C.c_char_p, C.c_int]),
    ("prth_make_sky", C.c_int, [C.c_int, C.c_int, C.c_void_p]),
    ("prth_hdr_load", C.c_void_p, [C.c_char_p, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.POINTER(C.c_float)), C.c_char_p, C.c_int]),
    ("prth_hdr_free", None, [C.c_void_p]),
    ("prth_hdr_write", C.c_int, [C.c_char_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_char_p, C.c_int]),
]

_lib = None


def load_library(path=None):
    """dlopen libprt.so and bind every prototype.  Raises if the library or a symbol is missing:
    there is no fallback path."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = path or LIB_PATH
    if not os.path.exists(p):
        raise RuntimeError("libprt.so not built: run `python __graft_entry__.py build` (%s)" % p)
    # PyTorch-ROCm ships its own copy of the HIP runtime.  Two copies in one process do not share the device: the one
    # that initialises second reports "no ROCm-capable device".  Loading torch FIRST makes libprt.so's libamdhip64
    # dependency resolve to the copy that is already there, so every Python process that might use both (bench.py,
    # build() followed by smoke(), tests that hand torch tensors to libprt) gets one runtime.
    try:
        import torch  # noqa: F401
    except Exception:
        pass
    lib = C.CDLL(p)
    for name, res, args in PRT_API + PRTH_API:
        fn = getattr(lib, name)          # AttributeError if the export is missing
        fn.restype = res
        fn.argtypes = args
    if path is None:
        _lib = lib
    return lib
