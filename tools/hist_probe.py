"""Batch ray rate against the hand-off threshold of the one-lane traversal (UPSP_HEAVY_STEPS_CAST), with the library's per-kernel
timers: one ray per pixel of a 1024^2 frame onto the frame-filling 1 M-triangle sphere (bench.py's pixel_rays_fill).
   python tools/hist_probe.py"""
import os, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) < 2:
    for h in (64, 96, 128, 256, 0):
        subprocess.call([sys.executable, __file__, "child"], env=dict(os.environ, UPSP_HEAVY_STEPS_CAST=str(h)))
    sys.exit(0)
import torch, bench
from upsp_processing_amd import _capi, engine, synthetic as syn
size = 1024
fv, ft = syn.cube_sphere(289, 6.0)
cd = syn.pinhole_camera(size, size, center=(0, 0, 20), half_extent=6.0, fill=0.95)
fs9, _ = syn.soup(fv, ft)
bvh = engine.BVH(fs9)
org, dirs = bench.pixel_rays(cd, size)
o, d = torch.as_tensor(org).cuda(), torch.as_tensor(dirs).cuda()
for _ in range(3):
    bvh.intersect(o, d, want=("hit", "t", "prim"))
torch.cuda.synchronize()
_capi.timing_enable(True)
for _ in range(5):
    bvh.intersect(o, d, want=("hit", "t", "prim"))
torch.cuda.synchronize()
_capi.timing_enable(False)
rep = _capi.timing_report()
print("hand-off past %3s steps:" % os.environ.get("UPSP_HEAVY_STEPS_CAST"), {k: round(v[1] / v[0], 4) for k, v in rep.items()}, flush=True)
