"""Registration sub-batches on the bench's own content (configs[2]: 1024^2 frames of the 1 M-triangle model with background,
fiducial discs and sub-pixel jitter: two ECC iterations per frame) for rocprofv3 --pmc passes over the ECC kernels."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from upsp_processing_amd import _capi, engine, synthetic as syn
size, F = 1024, int(os.environ.get("PROF_FRAMES", "128"))
verts, tris = syn.tunnel_model_quad()
s9, tn = syn.soup(verts, tris)
nrm = syn.node_normals(verts, tris)
N = verts.shape[0]
cd = syn.pinhole_camera(size, size, center=(0, 0, 20), half_extent=6.0, fill=0.7)
cam = _capi.make_camera(cd["K"], cd["dist"], cd["R"], cd["t"], size, size)
bvh = engine.BVH(s9)
d_nodes, d_nrm, d_tn = torch.as_tensor(verts).cuda(), torch.as_tensor(nrm).cuda(), torch.as_tensor(tn).cuda()
bvh.set_tri_nodes(d_tn, N)
proj = engine.build_projection(bvh, cam, d_nodes, d_nrm, d_tn, 70.0, counts=False)
layout = syn.scene_layout(proj["pix"].cpu().numpy(), size, size)
frames = torch.empty((F, size, size), dtype=torch.uint16, device="cuda")
for f0 in range(0, F, 50):
    syn.synth_frames_torch(min(50, F - f0), size, size, first=f0, out=frames[f0:f0 + 50], layout=layout, hot=True)
pipe = engine.FramePipeline(1, size, size, N, registration=1)
pipe.set_reference(0, frames[0].to(torch.float32))
pipe.set_projection(0, proj["pix"])
rows_t = torch.empty((N, engine.series_ld(F)), dtype=torch.float32, device="cuda")[:, :F]
for _ in range(2):
    pipe.reset()
    pipe.process(frames.clone(), 0, rows_t=rows_t, want_rows=False)
torch.cuda.synchronize()
print(pipe.ecc_stats())
