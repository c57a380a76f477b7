"""Lane utilisation of the traversal: closest-hit batch on the camera -> node rays of the bench model with statistics on."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from upsp_processing_amd import _capi, engine, synthetic as syn
size = 1024
verts, tris = syn.tunnel_model_quad()
s9, tn = syn.soup(verts, tris); nrm = syn.node_normals(verts, tris)
cd = syn.pinhole_camera(size, size, center=(0, 0, 20), half_extent=6.0)
cam = _capi.make_camera(cd["K"], cd["dist"], cd["R"], cd["t"], size, size)
bvh = engine.BVH(s9)
d_nodes, d_nrm, d_tn = [torch.as_tensor(x).cuda() for x in (verts, nrm, tn)]
bvh.set_tri_nodes(d_tn, verts.shape[0])
bvh.enable_stats(True)
cam_c = torch.tensor(engine.camera_center(cam), dtype=torch.float32, device="cuda")
d = d_nodes - cam_c
d = d / d.norm(dim=1, keepdim=True)
for _ in range(2):
    h = bvh.intersect(cam_c, d, want=("hit", "t", "prim"))
torch.cuda.synchronize()
s = bvh.last_stats()
print("closest-hit batch: nodes/ray %.1f tris/ray %.1f" % (s["nodes"] / s["rays"], s["tris"] / s["rays"]))
p = engine.build_projection(bvh, cam, d_nodes, d_nrm, d_tn, 70.0)
print("projection:", p["nrays"], bvh.last_stats())
