"""ctypes binding of oracle/libvrc_oracle.so -- the CPU restatement used as the
parity checker.  TEST INFRASTRUCTURE: imported only by tests/, by
__graft_entry__.smoke() and by bench.py's cpu_baseline leg."""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
LIB_PATH = os.path.join(ORACLE_DIR, "libvrc_oracle.so")

HIT_DTYPE = np.dtype([
    ("position", "<f4", 3), ("normal", "<f4", 3), ("voxel_coord", "<f4", 2),
    ("hit", "<u4"), ("node", "<u4"), ("distance", "<f4"), ("complexity", "<u4"),
])
assert HIT_DTYPE.itemsize == 48

LNODE_DTYPE = np.dtype([("color", "u1"), ("child_mask", "u1"), ("leaf_mask", "u1"),
                        ("pad", "u1"), ("child_offset", "<u4")])
assert LNODE_DTYPE.itemsize == 8


class Camera(C.Structure):
    _fields_ = [("position", C.c_float * 3), ("rot", C.c_float * 9),
                ("fov", C.c_float), ("aperture", C.c_float), ("focal_length", C.c_float)]


class FrameParams(C.Structure):
    _fields_ = [("width", C.c_uint32), ("height", C.c_uint32),
                ("light_position", C.c_float * 3),
                ("use_gi", C.c_uint32), ("use_samples", C.c_uint32),
                ("shadow_samples", C.c_uint32), ("gi_bounces", C.c_uint32),
                ("checker_parity", C.c_int32),
                ("seed", C.c_uint32), ("frame_index", C.c_uint32), ("rng_mode", C.c_uint32),
                ("row_block", C.c_uint32), ("shard_index", C.c_uint32), ("shard_count", C.c_uint32)]


class FrameStats(C.Structure):
    _fields_ = [("rays", C.c_uint64), ("sum_complexity", C.c_uint64), ("primary_hits", C.c_uint64)]


def build_oracle(force=False):
    src = os.path.join(ORACLE_DIR, "vrc_oracle.c")
    if (force or not os.path.exists(LIB_PATH)
            or os.path.getmtime(LIB_PATH) < os.path.getmtime(src)
            or os.path.getmtime(LIB_PATH) < os.path.getmtime(os.path.join(ORACLE_DIR, "vrc_oracle.h"))):
        subprocess.check_call(["make", "-C", ORACLE_DIR, "libvrc_oracle.so"], stdout=subprocess.DEVNULL)
    return LIB_PATH


GOLDEN = os.path.join(ROOT, "tests", "golden")


def load_terrain_heights():
    """int32 height[x, z] of main.cpp:69 for x, z < 1024: fixture generated from the reference's vendored FastNoise
    (oracle/_ref) by tests/tools/make_golden.py."""
    return np.load(os.path.join(GOLDEN, "terrain_height_1024.npz"))["height"].astype(np.int32)


def load_textures():
    """(top, side) 16x16 RGB tables decoded from the reference's res/grass_{top,side}_16x16.bmp (tests/golden)."""
    top = np.fromfile(os.path.join(GOLDEN, "grass_top_16x16.rgb"), np.uint8)
    side = np.fromfile(os.path.join(GOLDEN, "grass_side_16x16.rgb"), np.uint8)
    return top, side


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    build_oracle()
    L = C.CDLL(LIB_PATH)
    vp, u32, u64, f32, i32 = C.c_void_p, C.c_uint32, C.c_uint64, C.c_float, C.c_int32
    L.orc_svo_new.restype = vp
    L.orc_svo_new.argtypes = [u32]
    L.orc_svo_free.argtypes = [vp]
    L.orc_svo_set_cell.argtypes = [vp, u32, u32, u32]
    L.orc_svo_fill_terrain.argtypes = [vp, vp]
    L.orc_svo_compile.restype = u64
    L.orc_svo_compile.argtypes = [vp, C.POINTER(vp)]
    L.orc_free.argtypes = [vp]
    L.orc_cast_rays.argtypes = [vp, u32, u64, vp, vp, vp, vp, vp, C.c_int]
    L.orc_grid_cast_rays.argtypes = [vp, i32, i32, i32, u64, vp, vp, vp, C.c_int]
    L.orc_frac.restype = f32
    L.orc_frac.argtypes = [f32]
    L.orc_counter_rand.restype = u32
    L.orc_counter_rand.argtypes = [u32, u32, u32, u32]
    L.orc_get_rand_counter.restype = f32
    L.orc_get_rand_counter.argtypes = [u32, f32, f32]
    L.orc_xorshf96.restype = u64
    L.orc_xorshf96_fill.argtypes = [u64, vp]
    L.orc_make_rotation.argtypes = [f32, f32, vp]
    L.orc_render_frame.argtypes = [vp, u32, vp, vp, C.POINTER(Camera), C.POINTER(FrameParams),
                                   vp, vp, vp, C.POINTER(FrameStats), C.c_int]
    L.orc_samples_to_image.argtypes = [vp, u32, u32, vp]
    L.orc_terrain_heights.argtypes = [i32, i32, vp]
    _lib = L
    return L


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def compile_voxels(depth, voxels):
    """voxels: iterable of (x, y, z) -> LNode array via the restated SVO::setCell + compileSVO."""
    L = lib()
    s = L.orc_svo_new(depth)
    for (x, y, z) in voxels:
        L.orc_svo_set_cell(s, int(x), int(y), int(z))
    out = C.c_void_p()
    n = L.orc_svo_compile(s, C.byref(out))
    arr = np.frombuffer((C.c_uint8 * (n * 8)).from_address(out.value), dtype=LNODE_DTYPE).copy()
    L.orc_free(out)
    L.orc_svo_free(s)
    return arr


def compile_terrain(depth, height_i32):
    """height_i32[x, z] = main.cpp:69 `height`; returns the LNode array."""
    L = lib()
    size = 1 << depth
    h = np.ascontiguousarray(height_i32[:size, :size], dtype=np.int32)
    s = L.orc_svo_new(depth)
    L.orc_svo_fill_terrain(s, _ptr(h))
    out = C.c_void_p()
    n = L.orc_svo_compile(s, C.byref(out))
    arr = np.frombuffer((C.c_uint8 * (n * 8)).from_address(out.value), dtype=LNODE_DTYPE).copy()
    L.orc_free(out)
    L.orc_svo_free(s)
    return arr


def xorshf96_draws(n):
    """next n outputs of the reference's global xorshf96 stream (utils.cpp:11-25)"""
    out = np.zeros(n, np.uint64)
    lib().orc_xorshf96_fill(n, _ptr(out))
    return out


def get_rand(z, mn, mx):
    """getRand (utils.cpp:77-81) applied to raw xorshf96 outputs"""
    f = np.float32
    rv = (z % np.uint64(100)).astype(f) / f(100.0)
    return (f(mn) + (f(mx) - f(mn)) * rv).astype(f)


def terrain_heights(size, seed=1337):
    h = np.zeros((size, size), np.int32)
    lib().orc_terrain_heights(size, seed, _ptr(h))
    return h


def cast_rays(nodes, depth, org, dir_, coef=None, bias=None, threads=1):
    L = lib()
    org = np.ascontiguousarray(org, dtype=np.float32).reshape(-1, 3)
    dir_ = np.ascontiguousarray(dir_, dtype=np.float32).reshape(-1, 3)
    n = org.shape[0]
    out = np.zeros(n, dtype=HIT_DTYPE)
    if coef is not None:
        coef = np.ascontiguousarray(np.broadcast_to(np.asarray(coef, np.float32), (n,)))
    if bias is not None:
        bias = np.ascontiguousarray(np.broadcast_to(np.asarray(bias, np.float32), (n,)))
    L.orc_cast_rays(_ptr(nodes), depth, n, _ptr(org), _ptr(dir_), _ptr(coef), _ptr(bias), _ptr(out), threads)
    return out


def grid_cast_rays(cells, org, dir_, threads=1):
    L = lib()
    cells = np.ascontiguousarray(cells, dtype=np.uint8)
    X, Y, Z = cells.shape
    org = np.ascontiguousarray(org, dtype=np.float32).reshape(-1, 3)
    dir_ = np.ascontiguousarray(dir_, dtype=np.float32).reshape(-1, 3)
    out = np.zeros(org.shape[0], dtype=HIT_DTYPE)
    L.orc_grid_cast_rays(_ptr(cells), X, Y, Z, org.shape[0], _ptr(org), _ptr(dir_), _ptr(out), threads)
    return out


def make_rotation(angle_x, angle_y):
    rot = np.zeros(9, np.float32)
    lib().orc_make_rotation(angle_x, angle_y, _ptr(rot))
    return rot


def make_camera(position, rot, fov=1.0, aperture=0.0, focal_length=1.0):
    cam = Camera()
    cam.position[:] = [float(v) for v in position]
    cam.rot[:] = [float(v) for v in rot]
    cam.fov, cam.aperture, cam.focal_length = fov, aperture, focal_length
    return cam


def make_params(width, height, light, use_gi=0, use_samples=0, shadow_samples=0, gi_bounces=1,
                checker_parity=-1, seed=0x9E3779B9, frame_index=0, rng_mode=0,
                row_block=0, shard_index=0, shard_count=1):
    p = FrameParams()
    p.width, p.height = width, height
    p.light_position[:] = [float(v) for v in light]
    p.use_gi, p.use_samples, p.shadow_samples, p.gi_bounces = use_gi, use_samples, shadow_samples, gi_bounces
    p.checker_parity = checker_parity
    p.seed, p.frame_index, p.rng_mode = seed, frame_index, rng_mode
    p.row_block, p.shard_index, p.shard_count = row_block, shard_index, shard_count
    return p


def render_frame(nodes, depth, tex_top, tex_side, cam, params, image=None, accum=None,
                 want_prim=False, threads=1):
    L = lib()
    W, H = params.width, params.height
    if image is None:
        image = np.zeros((H, W, 4), np.uint8)
        image[..., 3] = 255
    if accum is None:
        accum = np.zeros((H, W, 4), np.uint32)
    prim = np.zeros(H * W, dtype=HIT_DTYPE) if want_prim else None
    st = FrameStats()
    L.orc_render_frame(_ptr(nodes), depth, _ptr(tex_top), _ptr(tex_side), C.byref(cam), C.byref(params),
                       _ptr(image), _ptr(accum), _ptr(prim), C.byref(st), threads)
    return image, accum, prim, st


def samples_to_image(accum):
    H, W, _ = accum.shape
    img = np.zeros((H, W, 4), np.uint8)
    lib().orc_samples_to_image(_ptr(accum), W, H, _ptr(img))
    return img


# ---- the reference's own thread pool (lib/swarm/swarm.hpp compiled where it lies into oracle/_ref/libref_swarm.so) ----

REF_SWARM = os.path.join(ORACLE_DIR, "_ref", "libref_swarm.so")


class AreaFrame(C.Structure):
    _fields_ = [("nodes", C.c_void_p), ("depth", C.c_uint32), ("tex_top", C.c_void_p), ("tex_side", C.c_void_p),
                ("cam", C.POINTER(Camera)), ("params", C.POINTER(FrameParams)), ("image", C.c_void_p), ("accum", C.c_void_p),
                ("rays", C.c_uint64 * 16), ("sum_complexity", C.c_uint64 * 16)]


class RefSwarm:
    """swrm::Swarm(16) of the reference, driving the oracle's per-area frame job (orc_area_job = the body of the swarm
    lambda, main.cpp:139-152).  Available where oracle/_ref/libref_swarm.so exists (built from /root/reference by
    `make -C oracle ref`; the .so travels to the GPU box, the reference's sources do not)."""

    def __init__(self, threads=16):
        self.lib = C.CDLL(REF_SWARM)
        self.lib.ref_swarm_create.restype = C.c_void_p
        self.lib.ref_swarm_create.argtypes = [C.c_uint32]
        self.lib.ref_swarm_destroy.argtypes = [C.c_void_p]
        self.lib.ref_swarm_run.restype = C.c_uint32
        self.lib.ref_swarm_run.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_uint32]
        self.threads = threads
        self.h = self.lib.ref_swarm_create(threads)
        self.job = C.cast(lib().orc_area_job, C.c_void_p)      # a plain C function pointer: no GIL on the workers
        self.retries = 0

    @staticmethod
    def available():
        return os.path.exists(REF_SWARM)

    def render_frame(self, nodes, depth, tex_top, tex_side, cam, params, image=None, accum=None):
        W, H = params.width, params.height
        if image is None:
            image = np.zeros((H, W, 4), np.uint8)
            image[..., 3] = 255
        if accum is None:
            accum = np.zeros((H, W, 4), np.uint32)
        f = AreaFrame()
        f.nodes, f.depth = _ptr(nodes), depth
        f.tex_top, f.tex_side = _ptr(tex_top), _ptr(tex_side)
        f.cam, f.params = C.pointer(cam), C.pointer(params)
        f.image, f.accum = _ptr(image), _ptr(accum)
        tries = self.lib.ref_swarm_run(self.h, self.threads, self.job, C.byref(f), 1000)
        if tries == 0:
            raise RuntimeError("the swarm never ran the job")
        self.retries += tries - 1
        st = FrameStats()
        st.rays, st.sum_complexity = sum(f.rays), sum(f.sum_complexity)
        return image, accum, st

    def close(self):
        if self.h:
            self.lib.ref_swarm_destroy(self.h)
            self.h = None
