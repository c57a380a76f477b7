"""Times the phase-2 kernel (delta-Cp from the node-major series) on the GPU.
    python tools/prof_phase2.py [--nodes N] [--frames F] [--reps R] [--inplace]
Algorithmic bytes: 4 B read + 4 B written per sample (pure streaming)."""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from upsp_processing_amd import engine  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--nodes", type=int, default=500958)
ap.add_argument("--frames", type=int, default=1000)
ap.add_argument("--reps", type=int, default=5)
ap.add_argument("--inplace", action="store_true")
a = ap.parse_args()
n, F = a.nodes, a.frames
g = torch.Generator(device="cuda").manual_seed(1)
I = 1500 + 40 * torch.randn((n, F), device="cuda", generator=g)
iref = I.mean(1)
cov = torch.ones(n, device="cuda")
cal = [1.2, -0.004, 1e-5, 0.02, 1e-4, -1e-7]
out = I.clone() if a.inplace else torch.empty_like(I)
src = out if a.inplace else I
for _ in range(2):
    engine.phase2_pressure(src, iref, cov, cal, 250.0, 1800.0, out=out)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(a.reps):
    engine.phase2_pressure(src, iref, cov, cal, 250.0, 1800.0, out=out)
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / a.reps
print("phase2 N=%d F=%d inplace=%d: %.3f ms  %.1f GB/s (8 B/sample)  %.2f Gsamples/s"
      % (n, F, a.inplace, ms, n * F * 8 / ms / 1e6, n * F / ms / 1e6))
