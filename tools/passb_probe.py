"""Pass B (upsp_rows_from_pixel_series) on the bench model's own node -> pixel table, alone on the device: the input series at
a 1024-frame pitch (what pass A writes) against a 1000-frame pitch (a block as it arrives from a peer: [pixel row][frames of the
source]), and warm (written a moment ago: Infinity Cache) against cold (2 GiB written in between).
Then the row padding (pad_to = 1024) and the owner's pass B of an 8-rank run: an eighth of the nodes, 8 blocks of 1000 frames into
rows of 8000 floats, launches cut at 128-byte lines of the rows against one launch per block (UPSP_ROWS_LINE_CUT=0).
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
    for cut in ("1", "0"):
        print("owner's pass B of 8 ranks, UPSP_ROWS_LINE_CUT=%s" % cut, flush=True)
        subprocess.call([sys.executable, __file__, "owner"], env=dict(os.environ, UPSP_ROWS_LINE_CUT=cut))
    sys.exit(0)


def blocks_call(bufs, pitches, counts, nk, sk, nn, out, ld, pad_to):
    m = len(bufs)
    ptrs = (C.c_void_p * m)(*[b.data_ptr() for b in bufs])
    _capi.check(L.upsp_rows_from_pixel_blocks(ptrs, (C.c_uint32 * m)(*pitches), (C.c_int64 * m)(*counts), m, C.c_void_p(nk.data_ptr()),
                                              C.c_void_p(sk.data_ptr()), nn, C.c_void_p(out.data_ptr()), ld, pad_to, C.c_void_p(s.data_ptr()),
                                              C.c_void_p(ss.data_ptr()), C.c_void_p(torch.cuda.current_stream().cuda_stream)))


def timed(run, cold, touch):
    ts = []
    for _ in range(6):
        junk.fill_(1.0) if cold else touch()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); run(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    return float(np.median(ts[1:])), min(ts[1:])


if sys.argv[1] == "owner":
    W = 8
    nn = N // W
    bl = [torch.randint(0, 4000, (A, F), generator=g, device="cuda", dtype=torch.int32).to(torch.uint16) for _ in range(W)]
    out = torch.empty((nn, W * F), dtype=torch.float32, device="cuda")
    for cold in (False, True):
        med, lo = timed(lambda: blocks_call(bl, [F] * W, [F] * W, d_nk[:nn], skipped[:nn], nn, out, W * F, W * F), cold,
                        lambda: [b.view(torch.int16).add_(0) for b in bl])
        print("  %d nodes x %d frames from %d blocks, %s: %.3f ms (min %.3f) = %.2f TB/s of row bytes" % (
            nn, W * F, W, "cold" if cold else "warm", med, lo, nn * W * F * 4 / med / 1e9), flush=True)
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
    for pad_to in (1000, 1024):
        for cold in (False, True):
            med, lo = timed(lambda: blocks_call([compact], [cp], [F], d_nk, skipped, N, rows, 1024, pad_to), cold,
                            lambda: compact.view(torch.int16).add_(0))
            print("series pitch %4d frames, %s, rows written to column %d: pass B %.3f ms (min %.3f) = %.2f TB/s of row bytes" % (
                cp, "cold" if cold else "warm", pad_to, med, lo, N * F * 4 / med / 1e9), flush=True)
