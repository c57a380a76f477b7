"""Pass B (upsp_rows_from_pixel_series) on the bench model's own node -> pixel table, alone on the device: the input series at
a 1024-frame pitch (what pass A writes) against a 1000-frame pitch (a block as it arrives from a peer: [pixel row][frames of the
source]), and warm (written a moment ago: Infinity Cache) against cold (2 GiB written in between).
   python tools/passb_probe.py"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from upsp_processing_amd import _capi, engine, synthetic as syn
size, F = 1024, 1000
verts, tris = syn.tunnel_model_quad()
s9, tn = syn.soup(verts, tris)
nrm = syn.node_normals(verts, tris)
N = verts.shape[0]
cd = syn.pinhole_camera(size, size, center=(0, 0, 20), half_extent=6.0, fill=0.7)
cam = _capi.make_camera(cd["K"], cd["dist"], cd["R"], cd["t"], size, size)
bvh = engine.BVH(s9)
d_nodes, d_nrm, d_tn = [torch.as_tensor(x).cuda() for x in (verts, nrm, tn)]
bvh.set_tri_nodes(d_tn, N)
pix = engine.build_projection(bvh, cam, d_nodes, d_nrm, d_tn, 70.0, counts=False)["pix"]
p = pix.cpu().numpy()
act = np.unique(p[p >= 0])
node_k = np.where(p >= 0, np.searchsorted(act, np.maximum(p, 0)), -1).astype(np.int32)
A = act.size
skipped = torch.as_tensor((p < 0).astype(np.uint8)).cuda()
d_nk = torch.as_tensor(node_k).cuda()
s = torch.zeros(N, dtype=torch.float64, device="cuda"); ss = torch.zeros_like(s)
rows = torch.empty((N, 1024), dtype=torch.float32, device="cuda")
junk = torch.empty(1 << 29, dtype=torch.float32, device="cuda")
L = _capi.lib()
g = torch.Generator(device="cuda"); g.manual_seed(1)
print("%d nodes, %d active pixels, %d nodes with a pixel" % (N, A, int((p >= 0).sum())))
import subprocess
if len(sys.argv) < 2:      # one child per kernel variant (the switch is read once per process)
    for var in ("40", "41", "42", "82", "162"):
        print("UPSP_ROWS_VARIANT=%s (sweeps per workgroup, series loads 0 = at use / 1 = one sweep ahead / 2 = all up front)" % var, flush=True)
        subprocess.call([sys.executable, __file__, "child"], env=dict(os.environ, UPSP_ROWS_VARIANT=var))
    sys.exit(0)
for cp in (1024, 1000):
    compact = torch.randint(0, 4000, (A, cp), generator=g, device="cuda", dtype=torch.int32).to(torch.uint16)
    def run():
        _capi.check(L.upsp_rows_from_pixel_series(C.c_void_p(compact.data_ptr()), cp, C.c_void_p(d_nk.data_ptr()), C.c_void_p(skipped.data_ptr()),
                                                  N, F, C.c_void_p(rows.data_ptr()), 1024, C.c_void_p(s.data_ptr()), C.c_void_p(ss.data_ptr()),
                                                  C.c_void_p(torch.cuda.current_stream().cuda_stream)))
    for cold in (False, True):
        ts = []
        for _ in range(6):
            if cold:
                junk.fill_(1.0)
            else:
                compact.view(torch.int16).add_(0)            # the series rewritten a moment ago
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); run(); e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        print("series pitch %4d frames, %s: pass B %.3f ms (min %.3f) = %.2f TB/s of row bytes" % (
            cp, "cold" if cold else "warm", float(np.median(ts[1:])), min(ts[1:]), N * F * 4 / float(np.median(ts[1:])) / 1e9), flush=True)
