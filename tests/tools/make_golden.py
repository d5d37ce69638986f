#!/usr/bin/env python3
"""Generate the committed fixtures under tests/golden/ from the reference's own
artefacts.  Runs ONLY where /root/reference is mounted (the dev container).

  terrain_height_1024.npz  int16 `height[x, z]` of src/main.cpp:69 for
                           x, z < 1024, produced by the REAL vendored FastNoise
                           (oracle/_ref/libref_fastnoise.so, compiled from
                           /root/reference/lib/fastnoise/FastNoise.cpp where it lies).
                           The generator evaluates noise(0.75x, 0.75z), which does
                           not depend on the scene size, so the top-left S x S
                           block is the S^3 scene's height field for S <= 1024.
  grass_top_16x16.rgb      768-byte top-down RGB tables decoded from the
  grass_side_16x16.rgb     reference data files res/grass_{top,side}_16x16.bmp
                           (24-bpp bottom-up BMP), i.e. what sf::Image::getPixel
                           returns (raycaster.hpp:53-54,239).

Fixtures are data (inputs / expected outputs), never reference source text.
"""
import ctypes as C
import os
import struct
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
REF = "/root/reference"
GOLD = os.path.join(ROOT, "tests", "golden")


def decode_bmp24(path):
    d = open(path, "rb").read()
    assert d[:2] == b"BM"
    off = struct.unpack_from("<I", d, 10)[0]
    w, h, planes, bpp, comp = struct.unpack_from("<iiHHI", d, 18)
    assert (w, h, bpp, comp) == (16, 16, 24, 0), (w, h, bpp, comp)
    stride = (w * 3 + 3) & ~3
    out = np.zeros((h, w, 3), np.uint8)
    for row in range(h):          # file row 0 = bottom image row
        y = h - 1 - row
        line = np.frombuffer(d, np.uint8, w * 3, off + row * stride).reshape(w, 3)
        out[y] = line[:, ::-1]    # BGR -> RGB
    return out


def main():
    if not os.path.isdir(REF):
        sys.exit("reference not mounted; fixtures can only be regenerated in the dev container")
    os.makedirs(GOLD, exist_ok=True)
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "ref"])
    lib = C.CDLL(os.path.join(ROOT, "oracle", "_ref", "libref_fastnoise.so"))
    size = 1024
    h = np.zeros((size, size), np.int32)
    lib.ref_terrain_heights(C.c_int32(size), h.ctypes.data_as(C.c_void_p))
    assert h.min() >= -32768 and h.max() <= 32767
    np.savez_compressed(os.path.join(GOLD, "terrain_height_1024.npz"), height=h.astype(np.int16))
    print("terrain heights", h.shape, "min", h.min(), "max", h.max())
    for name in ("grass_top_16x16", "grass_side_16x16"):
        img = decode_bmp24(os.path.join(REF, "res", name + ".bmp"))
        img.tofile(os.path.join(GOLD, name + ".rgb"))
        print(name, img.shape, img.reshape(-1, 3).mean(0))


if __name__ == "__main__":
    main()
