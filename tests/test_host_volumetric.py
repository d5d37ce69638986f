"""The drop-in class itself: `HipVolumetric : public Volumetric` (host/hip_raycaster.hpp) compiled against the reference's
own include/volumetric.hpp + include/cell.hpp and called through a `const Volumetric&` the way Camera::getClosestPoint
(include/camera_controller.hpp:56-60) calls it.  Round-2 verdict item 6.

GLM is not in this image.  <glm/glm.hpp> is satisfied by tests/standin_glm/glm/glm.hpp, a LABELLED STAND-IN that is not GLM:
this test pins nothing about the reference's numerics -- it only catches the day the class stops matching the interface it
claims to implement (a changed virtual signature, a missing override, a HitPoint field).  Runs only where /root/reference is
mounted (never on the GPU box)."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_INC = "/root/reference/include"
SRC = os.path.join(ROOT, "tests", "cpp", "hip_volumetric_main.cpp")
FLAGS = ["-std=c++14", "-Wall", "-Wextra", "-Werror=overloaded-virtual", "-I" + REF_INC, "-I" + os.path.join(ROOT, "tests", "standin_glm")]

pytestmark = pytest.mark.skipif(not os.path.isfile(os.path.join(REF_INC, "volumetric.hpp")),
                                reason="the reference's headers are not mounted here")


def test_hip_volumetric_is_a_volumetric_syntax_only():
    subprocess.run(["g++"] + FLAGS + ["-fsyntax-only", SRC], check=True)


def test_hip_volumetric_links_and_is_called_through_the_base_class(built, tmp_path):
    pkg = os.path.join(ROOT, "cpuvoxelraycaster_amd")
    exe = str(tmp_path / "hip_volumetric")
    subprocess.run(["g++"] + FLAGS + [SRC, "-o", exe, "-L" + pkg, "-l:libvrc_hip.so", "-Wl,-rpath," + pkg], check=True)
    out = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    # with a HIP device the ray is cast; without one the scene cannot be created: there is no CPU fallback behind the class
    assert out.stdout.startswith("castRay through Volumetric&") or "exception:" in out.stdout, out.stdout
