import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from upsp_processing_amd import _capi, engine, synthetic as syn
verts, tris = syn.tunnel_model_quad()
s9, tn = syn.soup(verts, tris)
bvh = engine.BVH(s9)
cam_c = torch.tensor([0, 0, 20.0], device="cuda")
d0 = torch.as_tensor(verts).cuda() - cam_c
for rep in (1, 2, 4, 8):
    d = d0.repeat(rep, 1).contiguous()
    for _ in range(2):
        bvh.intersect(cam_c, d, want=("t", "prim"))
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        bvh.intersect(cam_c, d, want=("t", "prim"))
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    print("rays %8d: %.3f ms  %.1f Mrays/s" % (d.shape[0], ms, d.shape[0] / ms / 1e3))
# interleaved copy (same rays adjacent) -> perfectly coherent pairs
