#!/usr/bin/env python3
"""Profiling driver: projection build (+ optional frame loop) on the bench model, with
traversal statistics.  Usage: python tools/prof_raycast.py [--reps N] [--frames F] [--stats]"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from upsp_processing_amd import _capi, engine, synthetic as syn

ap = argparse.ArgumentParser()
ap.add_argument("--reps", type=int, default=3)
ap.add_argument("--frames", type=int, default=0)
ap.add_argument("--stats", action="store_true")
ap.add_argument("--small", action="store_true")
ap.add_argument("--model", default="quad", choices=["quad", "uv"])
a = ap.parse_args()

size = 1024
if a.model == "uv":
    verts, tris = syn.tunnel_model(100, 240, 40, 80) if a.small else syn.tunnel_model()
else:
    verts, tris = syn.tunnel_model_quad(64, 24) if a.small else syn.tunnel_model_quad()
s9, tn = syn.soup(verts, tris)
nrm = syn.node_normals(verts, tris)
cd = syn.pinhole_camera(size, size, center=(0, 0, 20), half_extent=6.0)
cam = _capi.make_camera(cd["K"], cd["dist"], cd["R"], cd["t"], size, size)
bvh = engine.BVH(s9)
print("bvh", bvh.info)
d_nodes, d_nrm, d_tn = [torch.as_tensor(x).cuda() for x in (verts, nrm, tn)]
if a.stats:
    bvh.enable_stats(True)
for r in range(a.reps):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    p = engine.build_projection(bvh, cam, d_nodes, d_nrm, d_tn, 70.0)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    msg = "projection: %.3f ms, %d rays, %.1f Mrays/s, %d visible" % (
        dt * 1e3, p["nrays"], p["nrays"] / dt / 1e6, int((p["pix"] >= 0).sum()))
    if a.stats:
        s = bvh.last_stats()
        msg += " | nodes/ray %.1f tris/ray %.1f" % (s["nodes"] / max(s["rays"], 1), s["tris"] / max(s["rays"], 1))
    print(msg)
# plain closest-hit batch on camera->node rays (one ray per node, no retries)
cam_c = torch.tensor(engine.camera_center(cam), dtype=torch.float32, device="cuda")
d = d_nodes - cam_c
for r in range(a.reps):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    h = bvh.intersect(cam_c, d, want=("hit", "t", "prim"))
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    msg = "closest-hit: %.3f ms, %.1f Mrays/s" % (dt * 1e3, d.shape[0] / dt / 1e6)
    if a.stats:
        s = bvh.last_stats()
        msg += " | nodes/ray %.1f tris/ray %.1f" % (s["nodes"] / max(s["rays"], 1), s["tris"] / max(s["rays"], 1))
    print(msg)
if a.frames:
    F = a.frames
    frames = torch.empty((F, size, size), dtype=torch.uint16, device="cuda")
    for f0 in range(0, F, 50):
        syn.synth_frames_torch(min(50, F - f0), size, size, first=f0, out=frames[f0:f0 + 50])
    pipe = engine.FramePipeline(1, size, size, verts.shape[0])
    pipe.set_projection(0, p["pix"])
    rows_t = torch.empty((verts.shape[0], F), dtype=torch.float32, device="cuda")
    for r in range(a.reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        pipe.process(frames, 0, rows_t=rows_t, want_rows=False)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print("frame loop: %.3f ms for %d frames, %.0f frames/s" % (dt * 1e3, F, F / dt))
# per-kernel timing of one more projection build + closest-hit with normalised directions
_capi.timing_enable(True)
p = engine.build_projection(bvh, cam, d_nodes, d_nrm, d_tn, 70.0)
dn = d / d.norm(dim=1, keepdim=True)
h = bvh.intersect(cam_c, dn, want=("hit", "t", "prim"))
h = bvh.intersect(cam_c, d, want=("hit", "t", "prim"))
torch.cuda.synchronize()
for k, v in _capi.timing_report().items():
    print("timing", k, v)
print("hit fraction", h["hit"].float().mean().item())
