"""GPU: the header-only rt:: shim (include/upsp_rt.hpp) compiled with g++ against the C ABI --
the C++ call pattern of psp_process (cpp/exec/psp_process.cpp:44-53, 257-267)."""
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_rt_shim(gpu_lib, tmp_path):
    exe = str(tmp_path / "rt_shim_test")
    libdir = os.path.join(ROOT, "upsp_processing_amd", "lib")
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-I" + os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "cpp", "rt_shim_test.cpp"), "-o", exe,
                           "-L" + libdir, "-lupsp_gpu", "-Wl,-rpath," + libdir])
    out = subprocess.check_output([exe], text=True).split("\n")
    hit, t, x, y, z, prim = out[0].split()
    assert hit == "1" and abs(float(t) - 4.5) < 1e-6 and prim == "0"
    assert (float(x), float(y), float(z)) == (0.25, 0.25, 0.5)
    assert out[1].split()[0] == "0" and out[1].split()[2] == "-1"
    assert out[2].strip() == "1"
