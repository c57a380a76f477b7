#!/usr/bin/env python3
"""Frame-loop profiling driver: python tools/prof_frames.py --frames F [--ld LD]"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from upsp_processing_amd import _capi, engine, synthetic as syn
ap = argparse.ArgumentParser()
ap.add_argument("--frames", type=int, default=1000)
ap.add_argument("--ld", type=int, default=0)
ap.add_argument("--nodes", type=int, default=500958)
ap.add_argument("--vis", type=float, default=0.38)
ap.add_argument("--reps", type=int, default=3)
ap.add_argument("--size", type=int, default=1024)
ap.add_argument("--reg", action="store_true")
ap.add_argument("--filter", type=int, default=0)
ap.add_argument("--patch", type=int, default=0, help="number of fiducial clusters")
a = ap.parse_args()
size, N, F = a.size, a.nodes, a.frames
g = torch.Generator(device="cuda"); g.manual_seed(1)
# spatially coherent projection: node i -> pixel along a space-filling-ish sweep
pix = (torch.arange(N, device="cuda", dtype=torch.int64) * (size * size) // N).to(torch.int32)
pix[torch.rand(N, generator=g, device="cuda") > a.vis] = -1
frames = torch.empty((F, size, size), dtype=torch.uint16, device="cuda")
for f0 in range(0, F, 50):
    syn.synth_frames_torch(min(50, F - f0), size, size, first=f0, out=frames[f0:f0 + 50])
opts = {}
if a.reg: opts["registration"] = 1
if a.filter: opts.update(filter=1, filter_size=a.filter)
if a.patch: opts["patch"] = 1
pipe = engine.FramePipeline(1, size, size, N, **opts)
pipe.set_projection(0, pix)
pipe.set_reference(0, frames[0].to(torch.float32))
if a.patch:
    rng = np.random.default_rng(0)
    cl = []
    for k in range(a.patch):
        cx, cy = int(rng.integers(20, size - 20)), int(rng.integers(20, size - 20))
        xs, ys = np.meshgrid(np.arange(cx - 10, cx + 11), np.arange(cy - 10, cy + 11))
        r = np.hypot(xs - cx, ys - cy)
        b, i = (r > 5) & (r <= 8), r <= 5
        cl.append(dict(bx=xs[b], by=ys[b], ix=xs[i], iy=ys[i]))
    pipe.set_patches(0, cl)
ld = a.ld or F
buf = torch.empty((N, ld), dtype=torch.float32, device="cuda")
rows_t = buf[:, :F] if ld != F else buf
for r in range(a.reps):
    _capi.timing_enable(True)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    pipe.process(frames, 0, rows_t=rows_t, want_rows=False)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    rep = _capi.timing_report()
    print("F=%d ld=%d: %.3f ms  %.0f frames/s | " % (F, ld, dt * 1e3, F / dt) +
          " ".join("%s=%.3f(%d)" % (k.split("_kernel")[0], v[1], v[0]) for k, v in rep.items()))
