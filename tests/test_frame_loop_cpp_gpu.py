"""GPU: a torch-free C++ program drives the per-frame pipeline through the C ABI and the HIP runtime alone
(tests/cpp/frame_loop_test.cpp: createBVH -> create_projection_mat -> frame loop -> accumulators -> finals,
cpp/exec/psp_process.cpp:44-53, 167-355, 1771-1843, 1930-1936) and compares everything bit for bit with the golden
the oracle produced (tests/golden/make_golden_frame_loop.py)."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden", "frame_loop_sphere.bin")


@pytest.mark.gpu
def test_frame_loop_from_cpp(gpu_lib, tmp_path):
    exe = str(tmp_path / "frame_loop_test")
    libdir = os.path.join(ROOT, "upsp_processing_amd", "lib")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-O1", "-std=c++17", "-I" + os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "cpp", "frame_loop_test.cpp"), "-o", exe,
                           "-L" + libdir, "-lupsp_gpu", "-Wl,-rpath," + libdir])
    # no Python in the child: no torch, no numpy -- the library, the HIP runtime and libstdc++
    r = subprocess.run([exe, GOLDEN], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "frame loop ok: 9800 triangles, 4902 nodes, 1501 visible" in r.stdout, r.stdout
    ldd = subprocess.run(["ldd", exe], capture_output=True, text=True).stdout
    assert "torch" not in ldd and "python" not in ldd


def test_frame_loop_golden_is_what_the_oracle_computes(oracle):
    """The committed golden against a fresh run of its generator's oracle calls (header, projection, one row)."""
    import numpy as np
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    import make_golden_frame_loop as gen
    from upsp_processing_amd import synthetic as syn
    raw = open(GOLDEN, "rb").read()
    hdr = np.frombuffer(raw, np.uint32, 8)
    assert hdr[0] == 0x55505350 and tuple(hdr[1:6]) == (9800, 4902, 256, 256, 8)
    T, N, W, H, F = (int(x) for x in hdr[1:6])
    off = 32 + 8 * (9 + 5 + 9 + 3) + 4 + 4 * 3 * N * 2 + 4 * 3 * T
    pix = np.frombuffer(raw, np.int32, N, off)
    rows = np.frombuffer(raw, np.float32, F * N, off + 4 * N).reshape(F, N)
    v, t = syn.uv_sphere(50, 100)
    s9, tn = syn.soup(v, t)
    c = syn.pinhole_camera(W, H)
    cam = oracle.make_camera(c["K"], c["dist"], c["R"], c["t"], W, H)
    p = oracle.create_projection(oracle.OracleBVH(s9), cam, v, syn.node_normals(v, t), tn,
                                 np.float32((180.0 - 70.0) * 3.141592653589793 / 180.0))
    assert np.array_equal(p["pix"], pix) and int((pix >= 0).sum()) == hdr[6]
    frames = gen.make_frames()
    img, _ = oracle.fix_hot_pixels(frames[5])
    sol = oracle.project_frame(img, pix, None)
    ok = pix >= 0
    assert np.array_equal(sol[ok].view(np.int32), rows[5][ok].view(np.int32)) and np.isnan(rows[5][~ok]).all()
