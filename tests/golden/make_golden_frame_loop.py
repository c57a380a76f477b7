"""Golden vector of tests/cpp/frame_loop_test.cpp (a torch-free C++ caller of the per-frame pipeline): BASELINE
configs[0]'s 9 800-triangle sphere seen by a 256 x 256 pinhole camera, 8 frames of an integer pattern with hot pixels,
pushed through the ORACLE (oracle/: create_projection_mat, fix_hot_pixels, project_frame, NaN rows, double accumulators
-- cpp/exec/psp_process.cpp:167-355, 1771-1843).  Run once from the repository root:
    python tests/golden/make_golden_frame_loop.py
writes tests/golden/frame_loop_sphere.bin (little-endian, layout below).  The frames are NOT stored: both sides generate
them from the same integer hash (make_frames below = frame_value() of the C++ program)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import oracle as orc                      # noqa: E402
from upsp_processing_amd import synthetic as syn      # noqa: E402

W = H = 256
F = 8


def make_frames():
    """u16 [F, H, W]: integer ramp + hashed noise in 0..2047, folded below 3000; then a few hot pixels (>= 4064) -- frames 1
    and 5 get 2 and 5 of them (repaired), frame 3 gets 7 (more than max_hot = 5: left alone), frame 5's include image corners."""
    i = np.arange(H * W, dtype=np.uint64)
    y, x = np.divmod(i, W)
    fr = np.empty((F, H, W), np.uint16)
    M = np.uint64(0xFFFFFFFF)
    for f in range(F):
        h = (i * np.uint64(2654435761) + np.uint64(f * 40503 + 12345)) & M        # frame_value() of frame_loop_test.cpp
        h ^= h >> np.uint64(15)
        h = (h * np.uint64(2246822519)) & M
        h ^= h >> np.uint64(13)
        fr[f] = (((np.uint64(300) + np.uint64(5) * x + np.uint64(3) * y + np.uint64(11 * f) + (h >> np.uint64(21))) & np.uint64(0xFFF))
                 % np.uint64(3000)).astype(np.uint16).reshape(H, W)
    hot = {1: [(40, 50), (41, 50)], 3: [(10, 10), (20, 20), (30, 30), (40, 40), (50, 50), (60, 60), (70, 70)],
           5: [(0, 7), (100, 100), (100, 101), (200, 13), (255, 255)]}
    for f, lst in hot.items():
        for (yy, xx) in lst:
            fr[f, yy, xx] = 4095 - (yy % 16)
    return fr


def main():
    v, t = syn.uv_sphere(50, 100)                       # 9 800 triangles, 4 902 nodes
    s9, tn = syn.soup(v, t)
    nrm = syn.node_normals(v, t)
    c = syn.pinhole_camera(W, H)
    cam = orc.make_camera(c["K"], c["dist"], c["R"], c["t"], W, H)
    thr = np.float32((180.0 - 70.0) * 3.141592653589793 / 180.0)
    p = orc.create_projection(orc.OracleBVH(s9), cam, v, nrm, tn, thr)
    assert orc.oblique_ambiguous(cam, v, nrm, thr) == 0
    pix = p["pix"]
    sk = orc.skipped_nodes(pix)
    frames = make_frames()
    N = v.shape[0]
    rows = np.empty((F, N), np.float32)
    fixed = np.empty_like(frames)
    for f in range(F):
        img, _ = orc.fix_hot_pixels(frames[f])
        fixed[f] = img
        sol = orc.project_frame(img, pix, None)
        sol[sk] = np.nan
        rows[f] = sol
    assert (fixed[1] != frames[1]).sum() == 2 and (fixed[3] != frames[3]).sum() == 0 and (fixed[5] != frames[5]).sum() >= 4
    s = rows.astype(np.float64).sum(0)
    ss = (rows * rows).astype(np.float64).sum(0)       # sol * sol in float, accumulated in double (psp_process.cpp:1828-1831)
    dist = np.zeros(5, np.float64)
    dist[:len(np.ravel(c["dist"]))] = np.ravel(c["dist"])
    out = os.path.join(ROOT, "tests", "golden", "frame_loop_sphere.bin")
    with open(out, "wb") as fh:
        # header: magic, ntris, nnodes, W, H, F, visible nodes, frame checksum (sum of all repaired pixels mod 2^32)
        np.array([0x55505350, t.shape[0], N, W, H, F, int((pix >= 0).sum()), int(fixed.astype(np.uint64).sum() & 0xFFFFFFFF)],
                 np.uint32).tofile(fh)
        np.asarray(c["K"], np.float64).reshape(9).tofile(fh)
        dist.tofile(fh)
        np.asarray(c["R"], np.float64).reshape(9).tofile(fh)
        np.asarray(c["t"], np.float64).reshape(3).tofile(fh)
        np.array([thr], np.float32).tofile(fh)
        np.ascontiguousarray(v, np.float32).tofile(fh)
        np.ascontiguousarray(nrm, np.float32).tofile(fh)
        np.ascontiguousarray(t, np.int32).tofile(fh)
        pix.astype(np.int32).tofile(fh)
        rows.tofile(fh)
        s.tofile(fh)
        ss.tofile(fh)
    print("wrote", out, os.path.getsize(out), "bytes;", int((pix >= 0).sum()), "visible nodes of", N)


if __name__ == "__main__":
    main()
