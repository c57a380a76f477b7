"""What bounds projection_kernel<primary> after the early oblique test?  Builds on the bench model with the data-node
mask thinning the rays (random / contiguous), kernel times from the library's event timers."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from upsp_processing_amd import _capi, engine, synthetic as syn
size = 1024
verts, tris = syn.tunnel_model() if "uv" in sys.argv[1:] else syn.tunnel_model_quad()
s9, tn = syn.soup(verts, tris); nrm = syn.node_normals(verts, tris)
cd = syn.pinhole_camera(size, size, center=(0, 0, 20), half_extent=6.0)
cam = _capi.make_camera(cd["K"], cd["dist"], cd["R"], cd["t"], size, size)
bvh = engine.BVH(s9)
d_nodes, d_nrm, d_tn = [torch.as_tensor(x).cuda() for x in (verts, nrm, tn)]
bvh.set_tri_nodes(d_tn, verts.shape[0])
N = verts.shape[0]
rng = np.random.default_rng(1)
masks = {"all": None}
for f in (0.5, 0.1, 0.01):
    masks["random %.2f" % f] = (rng.random(N) < f).astype(np.uint8)
    m = np.zeros(N, np.uint8); m[: int(N * f)] = 1; masks["first %.2f" % f] = m
    m = np.zeros(N, np.uint8); m[N // 3: N // 3 + int(N * f)] = 1; masks["middle %.2f" % f] = m
for name, m in masks.items():
    dm = None if m is None else torch.as_tensor(m).cuda()
    for r in range(3):
        engine.build_projection(bvh, cam, d_nodes, d_nrm, d_tn, 70.0, datanode=dm, counts=False)
    torch.cuda.synchronize()
    _capi.timing_enable(True)
    for r in range(5):
        engine.build_projection(bvh, cam, d_nodes, d_nrm, d_tn, 70.0, datanode=dm, counts=False)
    torch.cuda.synchronize()
    _capi.timing_enable(False)
    rep = _capi.timing_report()
    pc = engine.projection_counts(bvh)
    print("%-12s primary rays %7d retry nodes %6d | " % (name, pc["primary_rays"], pc["retry_nodes"]) +
          "  ".join("%s %.1f us" % (k.replace("projection_kernel", "").replace("_kernel", "").replace("_kernels", ""), v[1] / v[0] * 1e3)
                    for k, v in rep.items() if v[0]), flush=True)
