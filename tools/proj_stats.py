import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from upsp_processing_amd import _capi, engine, synthetic as syn
v,t = syn.tunnel_model_quad()
s9,tn = syn.soup(v,t); nrm = syn.node_normals(v,t)
cd = syn.pinhole_camera(1024,1024, center=(0.2,0.1,20), half_extent=6.5)
cam = _capi.make_camera(cd["K"], cd["dist"], cd["R"], cd["t"], 1024, 1024)
bvh = engine.BVH(s9); d_tn = torch.as_tensor(tn).cuda()
dv, dn = torch.as_tensor(v).cuda(), torch.as_tensor(nrm).cuda()
bvh.enable_stats(True)
for mode in ("classic","bounded"):
    if mode=="bounded": bvh.set_tri_nodes(d_tn, v.shape[0])
    p = engine.build_projection(bvh, cam, dv, dn, d_tn, 70.0)
    st = bvh.last_stats()
    print(mode, st, "primary", p["primary_rays"], "retry_nodes", p["retry_nodes"], "per ray nodes %.1f tris %.1f" % (st["nodes"]/max(st["rays"],1), st["tris"]/max(st["rays"],1)))
