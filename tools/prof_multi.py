"""configs[4] shape on one GPU for rocprofv3 passes: 4 cameras, 5 M-triangle model, weighted frame loop of 512 frame sets
(pass A per camera + one whole-row pass B: node_rows_multi_kernel).  `small`: the 1 M-triangle model."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from upsp_processing_amd import _capi, engine, synthetic as syn
size, F, C = 1024, 512, 4
v, t = syn.tunnel_model_quad() if "small" in sys.argv[1:] else syn.tunnel_model_quad(576, 205)
s9, tn = syn.soup(v, t); nrm = syn.node_normals(v, t)
bvh = engine.BVH(s9)
dn, dm, dt = [torch.as_tensor(x).cuda() for x in (v, nrm, tn)]
bvh.set_tri_nodes(dt, v.shape[0])
cds = [syn.pinhole_camera(size, size, center=(0, 0, 20), half_extent=6.0, azimuth_deg=90.0 * c) for c in range(C)]
cams = [_capi.make_camera(cd["K"], cd["dist"], cd["R"], cd["t"], size, size) for cd in cds]
pix = torch.stack([engine.build_projection(bvh, cam, dn, dm, dt, 70.0, counts=False)["pix"] for cam in cams])
w = engine.projection_weights(pix, dn, dm, np.array([engine.camera_center(c) for c in cams]), "average_view")
frames = [syn.synth_frames_torch(F, size, size, first=100 * c, hot=True) for c in range(C)]
pipe = engine.FramePipeline(C, size, size, v.shape[0])
for c in range(C):
    pipe.set_projection(c, pix[c], w[c])
rows_t = torch.empty((v.shape[0], engine.series_ld(F, whole_rows=True)), dtype=torch.float32, device="cuda")[:, :F]
for r in range(2):
    pipe.reset()
    pipe.process(frames, 0, rows_t=rows_t, want_rows=False)
torch.cuda.synchronize()
print("done", v.shape[0], "nodes")
