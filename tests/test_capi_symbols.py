"""The C-ABI library loads and exports every symbol include/vrc.h declares; the
product fails loudly (no CPU fallback) when there is no GPU."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    txt = open(os.path.join(ROOT, "include", "vrc.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(vrc_[a-z0-9_]+)\s*\(", txt)))


def test_header_symbols_exported(built):
    from cpuvoxelraycaster_amd import capi
    lib = C.CDLL(capi.LIB_PATH)
    names = declared_symbols()
    assert len(names) >= 30
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/vrc.h but not exported"
    assert set(names) == set(capi.SYMBOLS), set(names) ^ set(capi.SYMBOLS)


def test_struct_sizes(built):
    from cpuvoxelraycaster_amd import capi
    assert C.sizeof(capi.Camera) == 15 * 4
    assert C.sizeof(capi.FrameParams) == 14 * 4
    assert C.sizeof(capi.FrameStats) == 32
    assert capi.HIT_DTYPE.itemsize == 48 and capi.LNODE_DTYPE.itemsize == 8


def test_host_cpp_mirror_compiles(built):
    """The C++ host adapter mirroring Volumetric / RayCaster compiles against the C ABI."""
    import subprocess
    hdr = os.path.join(ROOT, "cpuvoxelraycaster_amd", "host", "hip_raycaster.hpp")
    src = '#include "%s"\nint main(){ return 0; }\n' % hdr
    subprocess.run(["g++", "-std=c++14", "-fsyntax-only", "-x", "c++", "-"], input=src.encode(), check=True)


def test_rotation_helper_matches_oracle(built):
    import oracle_lib as O
    from cpuvoxelraycaster_amd import capi
    for ax, ay in [(0.0, 0.0), (0.0, -0.5), (0.3, -1.2), (-2.0, 0.7), (3.0, 1.5)]:
        assert np.array_equal(capi.make_rotation(ax, ay).view(np.uint32), O.make_rotation(ax, ay).view(np.uint32))


def test_no_gpu_fails_loudly(built):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    import cpuvoxelraycaster_amd as vrc
    nodes = np.zeros(9, vrc.LNODE_DTYPE)
    with pytest.raises(vrc.VrcError):
        vrc.LSVO(nodes, 2)
