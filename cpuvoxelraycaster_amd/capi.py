"""ctypes binding of libvrc_hip.so (the C ABI of include/vrc.h).

Plumbing only: every compute call goes to the hand-written HIP kernels.  There
is no CPU fallback -- if the library is missing, or no HIP device is present,
calls raise VrcError."""
import ctypes as C
import os

import numpy as np

PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(PKG, "libvrc_hip.so")

VRC_MEM_HOST, VRC_MEM_DEVICE = 0, 1

HIT_DTYPE = np.dtype([
    ("position", "<f4", 3), ("normal", "<f4", 3), ("voxel_coord", "<f4", 2),
    ("hit", "<u4"), ("node", "<u4"), ("distance", "<f4"), ("complexity", "<u4"),
])
LNODE_DTYPE = np.dtype([("color", "u1"), ("child_mask", "u1"), ("leaf_mask", "u1"),
                        ("pad", "u1"), ("child_offset", "<u4")])
assert HIT_DTYPE.itemsize == 48 and LNODE_DTYPE.itemsize == 8


class VrcError(RuntimeError):
    pass


class Camera(C.Structure):
    _fields_ = [("position", C.c_float * 3), ("rot", C.c_float * 9),
                ("fov", C.c_float), ("aperture", C.c_float), ("focal_length", C.c_float)]


class FrameParams(C.Structure):
    _fields_ = [("light_position", C.c_float * 3),
                ("use_gi", C.c_uint32), ("use_samples", C.c_uint32),
                ("shadow_samples", C.c_uint32), ("gi_bounces", C.c_uint32),
                ("checker_parity", C.c_int32), ("spp", C.c_uint32),
                ("seed", C.c_uint32), ("frame_index", C.c_uint32),
                ("row_block", C.c_uint32), ("shard_index", C.c_uint32), ("shard_count", C.c_uint32)]


class FrameStats(C.Structure):
    _fields_ = [("rays", C.c_uint64), ("sum_complexity", C.c_uint64),
                ("primary_hits", C.c_uint64), ("pixels", C.c_uint64), ("iterations_not_executed", C.c_uint64)]


# every symbol include/vrc.h declares: (restype, argtypes)
_vp, _u32, _u64, _i32, _f32, _int = C.c_void_p, C.c_uint32, C.c_uint64, C.c_int32, C.c_float, C.c_int
SYMBOLS = {
    "vrc_last_error": (C.c_char_p, []),
    "vrc_device_count": (_int, []),
    "vrc_scene_create": (_int, [_vp, _u64, _u32, _int, C.POINTER(_vp)]),
    "vrc_scene_set_textures": (_int, [_vp, _vp, _vp]),
    "vrc_scene_destroy": (_int, [_vp]),
    "vrc_scene_node_count": (_u64, [_vp]),
    "vrc_scene_depth": (_u32, [_vp]),
    "vrc_build_terrain_lsvo": (_int, [_vp, _u32, C.POINTER(_vp), C.POINTER(_u64)]),
    "vrc_build_volume_lsvo": (_int, [_vp, _u32, C.POINTER(_vp), C.POINTER(_u64)]),
    "vrc_free_host": (None, [_vp]),
    "vrc_scene_build_terrain": (_int, [_vp, _u32, _int, C.POINTER(_vp), C.POINTER(C.c_float)]),
    "vrc_scene_build_volume": (_int, [_vp, _u32, _int, C.POINTER(_vp), C.POINTER(C.c_float)]),
    "vrc_scene_download_nodes": (_int, [_vp, _vp]),
    "vrc_terrain_heights": (_int, [_i32, _u32, _int, _vp]),
    "vrc_scene_build_fastnoise_terrain": (_int, [_i32, _u32, _int, C.POINTER(_vp), C.POINTER(C.c_float)]),
    "vrc_cast_rays": (_int, [_vp, _u64, _vp, _vp, _vp, _vp, _vp, _int, _vp]),
    "vrc_cast_ray_chains": (_int, [_vp, _u64, _vp, _vp, _vp, _vp, C.c_float, _vp, _vp, _vp, _vp]),
    "vrc_cast_ray": (_int, [_vp, _vp, _vp, _f32, _f32, _vp]),
    "vrc_grid_create": (_int, [_vp, _i32, _i32, _i32, _int, C.POINTER(_vp)]),
    "vrc_grid_destroy": (_int, [_vp]),
    "vrc_grid_cast_rays": (_int, [_vp, _u64, _vp, _vp, _vp, _int, _vp]),
    "vrc_renderer_create": (_int, [_vp, _u32, _u32, C.POINTER(_vp)]),
    "vrc_renderer_destroy": (_int, [_vp]),
    "vrc_render_frame": (_int, [_vp, C.POINTER(Camera), C.POINTER(FrameParams), _vp]),
    "vrc_render_frame_resolved": (_int, [_vp, C.POINTER(Camera), C.POINTER(FrameParams), _vp, _vp]),
    "vrc_renderer_set_primary_capture": (_int, [_vp, _vp]),
    "vrc_samples_to_image": (_int, [_vp, _vp]),
    "vrc_reset_samples": (_int, [_vp, _vp]),
    "vrc_clear_image": (_int, [_vp, _vp]),
    "vrc_image_device_ptr": (_vp, [_vp]),
    "vrc_accum_device_ptr": (_vp, [_vp]),
    "vrc_read_image": (_int, [_vp, _vp, _vp]),
    "vrc_write_image": (_int, [_vp, _vp, _vp]),
    "vrc_read_accum": (_int, [_vp, _vp, _vp]),
    "vrc_get_stats": (_int, [_vp, C.POINTER(FrameStats), _int, _vp]),
    "vrc_shard_bytes": (_u64, [_u32, _u32, _u32, _u32]),
    "vrc_pack_shard": (_int, [_vp, _u32, _u32, _u32, _vp, _vp]),
    "vrc_resolve_shard": (_int, [_vp, _u32, _u32, _u32, _vp, _int, _vp]),
    "vrc_unpack_shards": (_int, [_vp, _u32, _u32, _u32, _u32, _vp, _vp]),
    "vrc_make_rotation": (None, [_f32, _f32, _vp]),
    "vrc_set_tuning": (_int, [_u32]),
    "vrc_presenter_create": (_int, [_int, _u32, _u32, _u32, _u32, C.POINTER(_vp)]),
    "vrc_presenter_destroy": (_int, [_vp]),
    "vrc_present": (_int, [_vp, _vp, _f32, _u32, _vp]),
    "vrc_present_image": (_int, [_vp, _vp, _f32, _u32, _vp]),
    "vrc_presenter_clear": (_int, [_vp, _vp]),
    "vrc_presenter_window_ptr": (_vp, [_vp]),
    "vrc_presenter_denoised_ptr": (_vp, [_vp]),
    "vrc_presenter_read": (_int, [_vp, _vp, _vp, _vp]),
    "vrc_selftest_exact_arith": (_int, [_int, _vp]),
    "vrc_set_sample_chunk": (_int, [_u32]),
    "vrc_renderer_set_tuning": (_int, [_vp, _u32]),
    "vrc_renderer_set_sample_chunk": (_int, [_vp, _u32]),
    "vrc_renderer_set_invariant_ray_reuse": (_int, [_vp, _u32]),
    "vrc_renderer_set_walk_from_root": (_int, [_vp, _u32]),
    "vrc_renderer_set_lane_samples": (_int, [_vp, _u32]),
    "vrc_set_lane_samples": (_int, [_u32]),
    "vrc_renderer_set_quad_walks": (_int, [_vp, _u32]),
    "vrc_renderer_last_kernel": (C.c_char_p, [_vp]),
    "vrc_ipc_export_image": (_int, [_vp, _vp]),
    "vrc_ipc_open_image": (_int, [_int, _vp, C.POINTER(_vp)]),
    "vrc_ipc_close_image": (_int, [_int, _vp]),
    "vrc_renderer_set_image_target": (_int, [_vp, _vp]),
    "vrc_ipc_flags_open": (_int, [C.c_char_p, _u32, _int, _int, C.POINTER(_vp)]),
    "vrc_ipc_flags_close": (_int, [_vp]),
    "vrc_ipc_flags_unlink": (_int, [_vp]),
    "vrc_ipc_stream_wait": (_int, [_vp, _vp, _vp, _u32, _u32]),
    "vrc_ipc_flag_set": (_int, [_vp, _u32, _u32]),
    "vrc_stream_write_flag": (_int, [_vp, _u32, _u32, _vp]),
    "vrc_stream_wait_flag": (_int, [_vp, _u32, _u32, _vp]),
    "vrc_ipc_flag_value": (_u32, [_vp, _u32]),
    "vrc_stream_create": (_int, [_int, C.POINTER(_vp)]),
    "vrc_stream_destroy": (_int, [_int, _vp]),
    "vrc_stream_synchronize": (_int, [_int, _vp]),
}

_lib = None


def load():
    """Load libvrc_hip.so; raises VrcError if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    try:
        # PyTorch bundles its own libamdhip64; it must be the HIP runtime of the
        # process (loading /opt/rocm's first makes torch see "No HIP GPUs").
        import torch  # noqa: F401
    except ImportError:
        pass
    path = LIB_PATH
    variant = os.environ.get("VRC_LIB", "").strip()
    if variant:
        # an EXPERIMENT build of the same sources (tools/build_variant.py -> gpurun_variants/var_<name>.so) loaded in the
        # product's place for an A/B -- never by overwriting the product library; said out loud, and bench.py records it
        if not os.path.exists(variant):
            raise VrcError(f"VRC_LIB={variant} does not exist")
        import sys
        print(f"cpuvoxelraycaster_amd: VRC_LIB set -- loading the experiment library {variant}, not the product {LIB_PATH}", file=sys.stderr)
        path = variant
    if not os.path.exists(path):
        raise VrcError(f"{path} not built: run `python -c 'import __graft_entry__ as g; g.build()'` "
                       "(the HIP library is the product; there is no fallback)")
    L = C.CDLL(path)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(L, name)
        fn.restype = res
        fn.argtypes = args
    _lib = L
    return L


def check(rc):
    if rc != 0:
        raise VrcError(f"vrc error {rc}: {load().vrc_last_error().decode()}")


def ptr(a):
    if a is None:
        return None
    if isinstance(a, np.ndarray):
        return a.ctypes.data_as(C.c_void_p)
    if isinstance(a, C.c_void_p):
        return a
    return C.c_void_p(int(a))  # raw device pointer (e.g. torch tensor .data_ptr())


def make_rotation(angle_x, angle_y):
    rot = np.zeros(9, np.float32)
    load().vrc_make_rotation(angle_x, angle_y, ptr(rot))
    return rot


def _take_host_nodes(out, n):
    arr = np.frombuffer((C.c_uint8 * (n.value * 8)).from_address(out.value), dtype=LNODE_DTYPE).copy()
    load().vrc_free_host(out)
    return arr


def build_terrain_lsvo(height_i32, depth):
    size = 1 << depth
    h = np.ascontiguousarray(np.asarray(height_i32)[:size, :size], dtype=np.int32)
    out, n = C.c_void_p(), C.c_uint64()
    check(load().vrc_build_terrain_lsvo(ptr(h), depth, C.byref(out), C.byref(n)))
    return _take_host_nodes(out, n)


def build_volume_lsvo(solid_u8, depth):
    size = 1 << depth
    s = np.ascontiguousarray(solid_u8, dtype=np.uint8)
    assert s.shape == (size, size, size)
    out, n = C.c_void_p(), C.c_uint64()
    check(load().vrc_build_volume_lsvo(ptr(s), depth, C.byref(out), C.byref(n)))
    return _take_host_nodes(out, n)
