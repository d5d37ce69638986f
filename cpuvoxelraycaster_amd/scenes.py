"""Synthetic inputs of the benchmark configurations:
the reference's terrain scene (src/main.cpp:59-76), camera (main.cpp:50-53)
and light (main.cpp:124-126), scaled by S/512 for other scene sizes."""
import os

import numpy as np

DATA = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data")


def terrain_heights(size=1024, seed=1337, device=0):
    """int32 height[x, z] of main.cpp:69 for x, z < size, evaluated on the GPU (vrc_terrain_heights: the reference's
    FastNoise settings, bit-identical to lib/fastnoise -- pinned by tests/test_gpu_builder.py against a fixture made
    with the reference's own FastNoise.cpp)."""
    from . import capi
    h = np.zeros((size, size), np.int32)
    capi.check(capi.load().vrc_terrain_heights(seed, size, device, capi.ptr(h)))
    return h


def load_textures():
    """(top, side) 16x16 RGB tables, top-down rows, as sf::Image::getPixel sees res/grass_{top,side}_16x16.bmp
    (raycaster.hpp:53-54): 768 bytes each, shipped with the package (data/)."""
    top = np.fromfile(os.path.join(DATA, "grass_top_16x16.rgb"), np.uint8)
    side = np.fromfile(os.path.join(DATA, "grass_side_16x16.rgb"), np.uint8)
    return top, side


def load_bmp(path):
    """An uncompressed 24- / 32-bpp BMP as sf::Image::loadFromFile + getPixel see it (raycaster.hpp:53-54,239): (h, w, 3)
    uint8 RGB, top row first.  For pointing the renderer at the reference's res/grass_{top,side}_16x16.bmp directly."""
    import struct
    d = open(path, "rb").read()
    if len(d) < 54 or d[:2] != b"BM":
        raise ValueError(f"{path}: not a BMP file")
    off = struct.unpack_from("<I", d, 10)[0]
    w, hs, _planes, bpp, comp = struct.unpack_from("<iiHHI", d, 18)
    h = abs(hs)
    if not (0 < w <= 16384 and 0 < h <= 16384) or bpp not in (24, 32) or not (comp == 0 or (comp == 3 and bpp == 32)):
        raise ValueError(f"{path}: only uncompressed 24 / 32 bpp is supported")
    bpx = bpp // 8
    stride = (w * bpx + 3) & ~3
    if len(d) < off + stride * h:
        raise ValueError(f"{path}: truncated file")
    out = np.zeros((h, w, 3), np.uint8)
    for row in range(h):
        y = row if hs < 0 else h - 1 - row
        line = np.frombuffer(d, np.uint8, w * bpx, off + row * stride).reshape(w, bpx)
        out[y] = line[:, 2::-1]                              # B, G, R [, A] -> R, G, B
    return out


def load_textures_bmp(top_path, side_path):
    """(top, side) tables for LSVO(..., textures=...) from two 16 x 16 BMP files."""
    top, side = load_bmp(top_path), load_bmp(side_path)
    if top.shape != (16, 16, 3) or side.shape != (16, 16, 3):
        raise ValueError("the albedo tables are 16 x 16")
    return top.reshape(-1), side.reshape(-1)


def reference_light(depth):
    """setLightPosition argument of main.cpp:124-126 for a 2^depth scene (SVO space)."""
    size = np.float32(1 << depth)
    world = np.array([-200.0, -1000.0, -300.0], np.float32) * np.float32(size / np.float32(512.0))
    return world * (np.float32(1.0) / size) + np.float32(1.0)


def reference_camera_position(depth):
    """Camera of main.cpp:50-53: (256, 200, 256) at 512^3, i.e. 56 voxels off the
    mid-plane y = S/2.  The terrain's thickness (16..89 voxels) does not scale
    with S, and the traversal sees the scene point-reflected (slot = idx ^
    mirror_mask, lsvo.hpp:79), so the solid band is y in [S/2 - lim, S/2 - 2]: the
    BASELINE rule 200*S/512 puts the camera INSIDE it for S < 512 (every primary
    ray then "hits" at t = 0 with an all-zero normal).  Keep the 56-voxel
    clearance for S <= 512 and the scaled height (free space) above."""
    s = 1 << depth
    y = s / 2 - 56.0 * max(1.0, s / 512.0)
    return (s / 2.0, max(y, 4.0), s / 2.0)


def reference_camera(depth, pitch=-0.5, yaw=0.0, aperture=0.0, focal_length=1.0, make_rotation=None):
    """Camera of main.cpp:50-53 (see reference_camera_position), fov 1."""
    from . import capi, raycaster
    rot = (make_rotation or capi.make_rotation)(yaw, pitch)
    return raycaster.make_camera(reference_camera_position(depth), rot, 1.0, aperture, focal_length)
