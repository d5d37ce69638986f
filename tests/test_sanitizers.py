"""AddressSanitizer + UndefinedBehaviorSanitizer on the CPU builds (GPU ASan is not available on the
pool): the oracle's whole surface and the product's host LSVO builder."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SAN = ["-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-ffp-contract=off"]


def test_oracle_under_asan_ubsan(tmp_path):
    exe = str(tmp_path / "san_oracle")
    subprocess.check_call(["gcc", *SAN, "-fopenmp", "-std=gnu99", "-I" + os.path.join(ROOT, "oracle"),
                           os.path.join(ROOT, "tests", "cpp", "sanitize_oracle_main.c"),
                           os.path.join(ROOT, "oracle", "vrc_oracle.c"), "-lm", "-o", exe])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "runtime error" not in out.stderr and "AddressSanitizer" not in out.stderr
    assert "nodes 666881" in out.stdout


def test_host_builder_under_asan_ubsan(tmp_path):
    exe = str(tmp_path / "san_builder")
    subprocess.check_call(["g++", *SAN, "-std=c++17", "-I" + os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "cpp", "sanitize_builder_main.cpp"),
                           os.path.join(ROOT, "cpuvoxelraycaster_amd", "csrc", "vrc_builder.cpp"), "-o", exe])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "runtime error" not in out.stderr and "AddressSanitizer" not in out.stderr
