"""Plain ray caster on the bench's pixel rays (one closest-hit ray per pixel of the 1024^2 frame): kernel times."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from upsp_processing_amd import _capi, engine, synthetic as syn
size = 1024
fill = "fill" in sys.argv[1:]
uv = "uv" in sys.argv[1:]
verts, tris = syn.cube_sphere(289, 6.0) if fill else syn.tunnel_model() if uv else syn.tunnel_model_quad()
s9, tn = syn.soup(verts, tris)
cd = syn.pinhole_camera(size, size, center=(0, 0, 20), half_extent=6.0)
bvh = engine.BVH(s9)
org, dirs = bench.pixel_rays(cd, size)
d_org, d_dirs = torch.as_tensor(org).cuda(), torch.as_tensor(dirs).cuda()
for want in (("hit", "t", "prim"), ("hit", "t", "prim", "uvw", "pos", "nrm")):
    for _ in range(3):
        h = bvh.intersect(d_org, d_dirs, want=want)
    torch.cuda.synchronize()
    _capi.timing_enable(True)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        h = bvh.intersect(d_org, d_dirs, want=want)
    e1.record()
    torch.cuda.synchronize()
    _capi.timing_enable(False)
    rep = _capi.timing_report()
    print("%s model, outputs %s: %.1f us per call; hit %.3f; " % ("fill" if fill else "uv" if uv else "tunnel", "+".join(want), e0.elapsed_time(e1) / 5 * 1e3, h["hit"].float().mean().item()) +
          "  ".join("%s %.1f us" % (k, v[1] / v[0] * 1e3) for k, v in rep.items() if v[0]), flush=True)
occ = bvh.occluded(d_org, d_dirs)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(5):
    occ = bvh.occluded(d_org, d_dirs)
e1.record(); torch.cuda.synchronize()
print("any-hit: %.1f us per call" % (e0.elapsed_time(e1) / 5 * 1e3))
