#!/usr/bin/env python3
"""VGPRs / SGPRs / spills / scratch of every kernel of csrc/vrc_kernels.hip, from the compiler's metadata (hipcc ... -S of the
device code with the product's flags + VRC_EXTRA_FLAGS).   usage: tools/code_objects.py [> profiles/rNN/code_objects.txt]"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cpuvoxelraycaster_amd import build  # noqa: E402

flags = [f for f in build.FLAGS if f not in ("-shared", "-fPIC")] + os.environ.get("VRC_EXTRA_FLAGS", "").split()
asm = subprocess.check_output([os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")] + flags +
                              ["--cuda-device-only", "-S", "-x", "hip", os.path.join(build.CSRC, "vrc_kernels.hip"), "-o", "-"], text=True)
rows = []
for blk in asm.split("  - .agpr_count:")[1:]:
    g = lambda k: int(re.search(r"\." + k + r":\s+(\d+)", blk).group(1))
    name = re.search(r"\.name:\s+(\S+)", blk).group(1)
    rows.append((name, g("vgpr_count"), g("vgpr_spill_count"), g("sgpr_count"), g("sgpr_spill_count"), g("private_segment_fixed_size")))
print(f"{'kernel':44s} {'VGPRs':>6s} {'VGPR spills':>12s} {'SGPRs':>6s} {'SGPR spills':>12s} {'scratch B/lane':>15s}")
for r in sorted(rows):
    print(f"{r[0]:44s} {r[1]:6d} {r[2]:12d} {r[3]:6d} {r[4]:12d} {r[5]:15d}")
