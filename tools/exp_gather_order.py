#!/usr/bin/env python3
"""Experiment: how much does the node order (pixel locality of adjacent lanes) matter for the
gather?  Same projection, nodes in mesh order / sorted by pixel / randomly permuted."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from upsp_processing_amd import _capi, engine, synthetic as syn
size, F = 1024, 1024
verts, tris = syn.tunnel_model_quad()
s9, tn = syn.soup(verts, tris); nrm = syn.node_normals(verts, tris); N = verts.shape[0]
cd = syn.pinhole_camera(size, size, center=(0, 0, 20), half_extent=6.0)
cam = _capi.make_camera(cd["K"], cd["dist"], cd["R"], cd["t"], size, size)
bvh = engine.BVH(s9)
d_nodes, d_nrm, d_tn = (torch.as_tensor(x).cuda() for x in (verts, nrm, tn))
bvh.set_tri_nodes(d_tn, N)
pix = engine.build_projection(bvh, cam, d_nodes, d_nrm, d_tn, 70.0, counts=False)["pix"]
frames = torch.empty((F, size, size), dtype=torch.uint16, device="cuda")
for f0 in range(0, F, 64):
    syn.synth_frames_torch(64, size, size, first=f0, out=frames[f0:f0 + 64])
ld = engine.series_ld(F)
buf = torch.empty((N, ld), dtype=torch.float32, device="cuda")[:, :F]
vis = pix >= 0
print("visible", int(vis.sum()), "of", N)
# tiles (64 nodes) by number of visible nodes
tv = torch.nn.functional.pad(vis, (0, (-N) % 64)).view(-1, 64).sum(1)
print("tiles: all-invisible %d, full %d, partial %d" % (int((tv == 0).sum()), int((tv == 64).sum()), int(((tv > 0) & (tv < 64)).sum())))
order_sorted = torch.argsort(torch.where(vis, pix, torch.full_like(pix, 1 << 30)), stable=True)
g = torch.Generator(device="cuda"); g.manual_seed(0)
order_vis = torch.argsort((~vis).to(torch.int8), stable=True)          # visible nodes first, mesh order kept
variants = {"mesh order": pix, "visible first": pix[order_vis], "sorted by pixel": pix[order_sorted],
            "random order": pix[torch.randperm(N, device="cuda", generator=g)]}
for name, p in variants.items():
    pipe = engine.FramePipeline(1, size, size, N)
    pipe.set_projection(0, p.contiguous())
    for r in range(3):
        _capi.timing_enable(True)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        pipe.process(frames, 0, rows_t=buf, want_rows=False)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        rep = _capi.timing_report()
    print("%-16s %.3f ms | " % (name, dt * 1e3) + " ".join("%s=%.1fus(%d)" % (k.split("_kernel")[0], 1e3 * v[1] / v[0], v[0]) for k, v in rep.items()))
