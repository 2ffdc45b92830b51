"""The C ABI from plain C: tests/c_abi/host_smoke.c is compiled with gcc against include/m2d.h, linked to
foodrec_amd/libm2d.so and the HIP runtime, and run as its own process (no Python, no torch in it)."""
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_c_host_program(tmp_path):
    exe = str(tmp_path / "host_smoke")
    lib = os.path.join(ROOT, "foodrec_amd")
    cmd = ["gcc", "-O1", "-std=gnu11", os.path.join(ROOT, "tests", "c_abi", "host_smoke.c"), "-I" + os.path.join(ROOT, "include"),
           "-I/opt/rocm/include", "-L" + lib, "-L/opt/rocm/lib", "-lm2d", "-lamdhip64", "-lm",
           "-Wl,-rpath," + lib + ",-rpath,/opt/rocm/lib", "-o", exe]
    subprocess.run(cmd, check=True, capture_output=True, text=True)
    res = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert res.returncode == 0, res.stdout + res.stderr
    assert "C ABI OK" in res.stdout and "KAT score = 3.4625" in res.stdout
