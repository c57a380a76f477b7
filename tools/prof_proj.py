"""Projection builds on the bench model (adjacency handed over) for rocprofv3 --pmc passes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from upsp_processing_amd import _capi, engine, synthetic as syn
size = 1024
verts, tris = syn.tunnel_model() if 'uv' in sys.argv[1:] else syn.tunnel_model_quad()
s9, tn = syn.soup(verts, tris); nrm = syn.node_normals(verts, tris)
cd = syn.pinhole_camera(size, size, center=(0, 0, 20), half_extent=6.0)
cam = _capi.make_camera(cd["K"], cd["dist"], cd["R"], cd["t"], size, size)
bvh = engine.BVH(s9)
d_nodes, d_nrm, d_tn = [torch.as_tensor(x).cuda() for x in (verts, nrm, tn)]
bvh.set_tri_nodes(d_tn, verts.shape[0])
for r in range(3):
    p = engine.build_projection(bvh, cam, d_nodes, d_nrm, d_tn, 70.0, counts="ref" in sys.argv[1:])
    if "nrays" not in p: p.update(engine.projection_counts(bvh))
torch.cuda.synchronize()
print(p["nrays"], p["primary_rays"], p["retry_nodes"])
