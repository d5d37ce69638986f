"""The documented experiment builds still compile (round-5 advice: quad_waves() named a macro that only the product build
defined, so every -DVRC_PROBE / -DVRC_WALK_CPP / -DVRC_NO_START_BELOW build was broken without anyone noticing).  A front-end
pass (hipcc -fsyntax-only: host AND gfx950 device side, inline-asm constraints included) of the two sources that carry
switches, once per flag of cpuvoxelraycaster_amd/build.py's VARIANT_FLAGS; about a second each, no GPU."""
import os
import subprocess

import pytest

from cpuvoxelraycaster_amd import build

HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FRONT = [f for f in build.FLAGS if f not in ("-shared", "-fPIC", "-O3")] + ["-fsyntax-only", "-Werror=implicit-function-declaration"]


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="no hipcc")
@pytest.mark.parametrize("flag", [""] + build.VARIANT_FLAGS)
def test_variant_flag_compiles(flag):
    for src in ("vrc_kernels.hip", "vrc_api.cpp"):
        cmd = [HIPCC] + FRONT + ([flag] if flag else []) + ["-x", "hip", os.path.join(build.CSRC, src)]
        r = subprocess.run(cmd, capture_output=True, text=True)
        assert r.returncode == 0, f"{' '.join(cmd)}\n{r.stderr[-3000:]}"


def test_variant_builds_have_no_quadrant_kernels_and_say_so():
    """the variants that leave the quadrant-walk kernels out must report it (vrc::quad_available), so that render_impl neither
    sets FrameArgs::quad_walks nor sizes the grid for a build launch_render would not launch"""
    src = open(os.path.join(build.CSRC, "vrc_kernels.hip")).read()
    api = open(os.path.join(build.CSRC, "vrc_api.cpp")).read()
    assert "bool quad_available()" in src and "vrc::quad_available() &&" in api
    # the occupancy macro is defined whatever the variant
    head = src[:src.index("#define VRC_HAVE_QUAD 1")]
    assert "#define VRC_SYNC_Q_WAVES 6" in head and head.rstrip().endswith("defined(VRC_NO_START_BELOW))")


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="no hipcc")
def test_frame_kernels_keep_their_register_budgets_and_the_committed_record_is_current():
    """round-5 advice: the quadrant-walk kernels sit at the 80-VGPR budget of 6 waves per SIMD; scratch traffic inside the unit
    loop of the timed kernel would cost time without failing any parity test.  The compiler's metadata of THIS tree (device
    side of csrc/vrc_kernels.hip, product flags) against the accepted spills of tools/code_objects.py, and against the newest
    committed profiles/rNN/code_objects.txt (what DESIGN.md and the bench notes quote)."""
    import glob
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "tools"))
    import code_objects as co
    rows = co.kernel_rows(extra_flags=[])
    by_name = {r[0]: r for r in rows}
    for name, (spills, scratch) in co.ACCEPTED_SPILLS.items():
        _, vgprs, vspill, _, _, scr = by_name[name]
        assert vspill <= spills and scr <= scratch, (name, vspill, scr, "accepted", spills, scratch)
        assert vgprs <= (72 if name.endswith("_w7") else 80), (name, vgprs)          # 7 / 6 waves per SIMD of 512 VGPRs
    # no frame kernel outside the table spills more than the lens kernels do
    assert all(r[5] <= 72 for r in rows if r[0].startswith("k_render_sync")), [r for r in rows if r[5] > 72]
    newest = sorted(glob.glob(os.path.join(root, "profiles", "r[0-9][0-9]", "code_objects.txt")))[-1]
    assert open(newest).read() == co.table(rows), f"{newest} is not this tree's compiler output: python tools/code_objects.py > {newest}"
