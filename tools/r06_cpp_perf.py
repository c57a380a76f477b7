#!/usr/bin/env python3
"""Round 6: the one-call re-raycast step driven from C++ (tests/cpp/frame_loop_test.cpp --perf) under a few environment switches.
  python3 tools/r06_cpp_perf.py [VAR=value ...]   (each argument = one extra run with that assignment; always runs the default first)"""
import os
import struct
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
from upsp_processing_amd import engine, synthetic as syn  # noqa: E402


def main():
    tmp = tempfile.mkdtemp()
    verts, tris = syn.tunnel_model_quad()
    _, tn = syn.soup(verts, tris)
    nrm = syn.node_normals(verts, tris)
    size = 1024
    c = syn.pinhole_camera(size, size, center=(0, 0, 20), half_extent=6.0, fill=0.7)
    path = os.path.join(tmp, "model.bin")
    with open(path, "wb") as f:
        f.write(struct.pack("<8I", 0x5550534D, tris.shape[0], verts.shape[0], size, size, 0, 0, 0))
        for k in ("K", "dist", "R", "t"):
            f.write(np.asarray(c[k], np.float64).tobytes())
        f.write(struct.pack("<f", engine.oblique_threshold(70.0)))
        f.write(np.ascontiguousarray(verts, np.float32).tobytes())
        f.write(np.ascontiguousarray(nrm, np.float32).tobytes())
        f.write(np.ascontiguousarray(tn, np.int32).tobytes())
    exe = os.path.join(tmp, "frame_loop_test")
    libdir = os.path.join(ROOT, "upsp_processing_amd", "lib")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-O1", "-std=c++17", "-I" + os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "cpp", "frame_loop_test.cpp"), "-o", exe, "-L" + libdir, "-lupsp_gpu", "-Wl,-rpath," + libdir])
    for rep in range(2):
        for kv in [""] + sys.argv[1:]:
            env = dict(os.environ)
            for a in kv.split(","):
                if a:
                    k, v = a.split("=", 1)
                    env[k] = v
            r = subprocess.run([exe, "--perf", path, "1000", "40"], capture_output=True, text=True, env=env, timeout=600)
            print("%-40s %s" % (kv or "(default)", (r.stdout.strip() or r.stderr.strip())[:160]), flush=True)


if __name__ == "__main__":
    main()
