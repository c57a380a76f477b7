#!/usr/bin/env python3
"""Debug: steps-per-ray histogram of the residual retry traversal on the bench model
(UPSP_DEBUG_HIST=1 UPSP_DEBUG_COUNTS=1 python tools/residual_hist.py)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from upsp_processing_amd import _capi, engine, synthetic as syn
v, t = syn.tunnel_model_quad()
s9, tn = syn.soup(v, t); nrm = syn.node_normals(v, t)
cd = syn.pinhole_camera(1024, 1024, center=(0, 0, 20), half_extent=6.0)
cam = _capi.make_camera(cd["K"], cd["dist"], cd["R"], cd["t"], 1024, 1024)
bvh = engine.BVH(s9); d_tn = torch.as_tensor(tn).cuda()
dv, dn = torch.as_tensor(v).cuda(), torch.as_tensor(nrm).cuda()
bvh.set_tri_nodes(d_tn, v.shape[0])
if os.environ.get("UPSP_DEBUG_HIST"): bvh.enable_stats(True)
p = engine.build_projection(bvh, cam, dv, dn, d_tn, 70.0)
print(bvh.last_stats(), p["primary_rays"], p["retry_nodes"])
