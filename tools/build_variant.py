#!/usr/bin/env python3
"""Builds libvrc_hip.so variants HERE (hipcc cross-compiles) into gpurun_variants/var_<name>.so, so that an A/B on the GPU
box (tools/ab_prebuilt.sh) spends no box time compiling.   usage: tools/build_variant.py name="-DFLAG ..." [name2=...]"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cpuvoxelraycaster_amd import build  # noqa: E402

OUT = os.path.join(ROOT, "gpurun_variants")


def one(arg):
    name, _, flags = arg.partition("=")
    lib = os.path.join(OUT, f"var_{name}.so")
    cmd = [os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")] + build.FLAGS + flags.split() + ["-x", "hip"] + \
          [os.path.join(build.CSRC, s) for s in build.SOURCES] + ["-o", lib]
    subprocess.check_call(cmd)
    with open(lib + ".flags", "w") as f:
        f.write(flags + "\n")
    return lib


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    with ThreadPoolExecutor(max_workers=4) as ex:
        for lib in ex.map(one, sys.argv[1:]):
            print(lib)
