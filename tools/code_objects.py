#!/usr/bin/env python3
"""VGPRs / SGPRs / spills / scratch of every kernel of csrc/vrc_kernels.hip, from the compiler's metadata (hipcc ... -S of the
device code with the product's flags + VRC_EXTRA_FLAGS).   usage: tools/code_objects.py [> profiles/rNN/code_objects.txt]

tests/test_variant_builds.py holds the frame kernels to the budgets of ACCEPTED_SPILLS below and the newest committed
profiles/rNN/code_objects.txt to this tree's compiler output."""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cpuvoxelraycaster_amd import build  # noqa: E402

# kernel -> (VGPR spills, scratch bytes per lane) the shipped builds are accepted with (round-5 advice: the quadrant-walk
# kernels sit at the 80-VGPR budget; a spill that creeps into the unit loop of the timed kernel is a performance regression
# nobody would see in the parity tests).  The one spill of k_render_sync_pinhole_resolved_q is the fused resolve's tile counter
# pointer, touched once per unit; the 2-bounce extension's five are in the second gi_chain.  Everything else of the pinhole
# family has none.  The lens kernels' figures are the accepted cost of 7 waves per SIMD.
ACCEPTED_SPILLS = {
    "k_render_sync_pinhole_q": (0, 0),
    "k_render_sync_pinhole_resolved_q": (1, 8),
    "k_render_sync_pinhole2_q": (5, 24),
    "k_render_sync_pinhole2_resolved_q": (5, 24),
    "k_render_sync_pinhole": (0, 0),
    "k_render_sync_pinhole_resolved": (0, 0),
    "k_render_sync_w7": (14, 60),
    "k_render_sync_resolved_w7": (17, 72),
}


def kernel_rows(extra_flags=None):
    """[(name, VGPRs, VGPR spills, SGPRs, SGPR spills, scratch bytes per lane)] of every kernel, sorted by name"""
    flags = [f for f in build.FLAGS if f not in ("-shared", "-fPIC")] + (os.environ.get("VRC_EXTRA_FLAGS", "").split() if extra_flags is None else list(extra_flags))
    asm = subprocess.check_output([os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")] + flags +
                                  ["--cuda-device-only", "-S", "-x", "hip", os.path.join(build.CSRC, "vrc_kernels.hip"), "-o", "-"],
                                  text=True, stderr=subprocess.DEVNULL)
    rows = []
    for blk in asm.split("  - .agpr_count:")[1:]:
        g = lambda k: int(re.search(r"\." + k + r":\s+(\d+)", blk).group(1))
        name = re.search(r"\.name:\s+(\S+)", blk).group(1)
        rows.append((name, g("vgpr_count"), g("vgpr_spill_count"), g("sgpr_count"), g("sgpr_spill_count"), g("private_segment_fixed_size")))
    return sorted(rows)


def table(rows):
    out = [f"{'kernel':44s} {'VGPRs':>6s} {'VGPR spills':>12s} {'SGPRs':>6s} {'SGPR spills':>12s} {'scratch B/lane':>15s}"]
    out += [f"{r[0]:44s} {r[1]:6d} {r[2]:12d} {r[3]:6d} {r[4]:12d} {r[5]:15d}" for r in rows]
    return "\n".join(out) + "\n"


if __name__ == "__main__":
    sys.stdout.write(table(kernel_rows()))
