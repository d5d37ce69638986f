"""Builds libvrc_hip.so (HIP kernels + C ABI) in-tree for gfx950 with hipcc."""
import os
import subprocess

PKG = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG, "csrc")
LIB = os.path.join(PKG, "libvrc_hip.so")
SOURCES = ["vrc_kernels.hip", "vrc_build_gpu.hip", "vrc_present.hip", "vrc_api.cpp", "vrc_builder.cpp"]
HEADERS = ["vrc_device.h", "vrc_walk.h", "vrc_internal.h", os.path.join("..", "..", "include", "vrc.h")]

# -ffp-contract=off: hipcc fuses a*b+c into v_fma_f32 by default, which changes
# which voxel a ray hits.  Correctly rounded f32 divide and
# sqrt are hipcc's default; stated explicitly because parity depends on them.
# -fno-slp-vectorize: hipcc packs pairs of scalar f32 operations into v_pk_* and keeps the pairs live, which costs a
# third more register spills in the frame kernels for nothing (a packed op issues at the cost of two): 0.6-4 % on every
# BASELINE configuration (tools/ab_variants.sh).
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
         "-ffp-contract=off", "-fhip-fp32-correctly-rounded-divide-sqrt",
         "-fno-fast-math", "-fno-slp-vectorize", "-Wall", "-Wno-unused-result"]


# The experiment switches of the sources (VRC_EXTRA_FLAGS / tools/build_variant.py / tools/probe_waves.py); never part of the
# product's flag line.  tests/test_variant_builds.py holds every one of them to a compiler front-end pass (host and device).
VARIANT_FLAGS = ["-DVRC_PROBE", "-DVRC_WALK_CPP", "-DVRC_NO_START_BELOW", "-DVRC_NO_PINHOLE_KERNEL", "-DVRC_WALK_LATE_WAIT",
                 "-DVRC_WALK_GLOBAL_LOAD", "-DVRC_EXP_UNITS", "-DVRC_SYNC_Q_WAVES=7", "-DVRC_SYNC_MIN_WAVES=5"]


STAMP = LIB + ".flags"   # the flags the library was built with: an experiment build is never mistaken for the product


def _flag_line():
    return " ".join(FLAGS + os.environ.get("VRC_EXTRA_FLAGS", "").split())


def needs_build():
    if not os.path.exists(LIB) or not os.path.exists(STAMP):
        return True
    with open(STAMP) as f:
        if f.read().strip() != _flag_line():
            return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in SOURCES + HEADERS] + [os.path.abspath(__file__)]
    return any(os.path.getmtime(d) > t for d in deps)


def build_native(force=False, verbose=False):
    if not force and not needs_build():
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    extra = os.environ.get("VRC_EXTRA_FLAGS", "").split()   # experiments only
    cmd = [hipcc] + FLAGS + extra + ["-x", "hip"] + [os.path.join(CSRC, s) for s in SOURCES] + ["-o", LIB]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    with open(STAMP, "w") as f:
        f.write(_flag_line() + "\n")
    return LIB


if __name__ == "__main__":
    print(build_native(force=True, verbose=True))
