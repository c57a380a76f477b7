"""Host-side time of every call of the chunked (N > 1 shaped) frame loop on one GPU."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from upsp_processing_amd import _capi, engine, synthetic as syn, distributed as D
size, F = 1024, 1000
verts, tris = syn.tunnel_model_quad()
s9, tn = syn.soup(verts, tris); nrm = syn.node_normals(verts, tris); N = verts.shape[0]
cd = syn.pinhole_camera(size, size, center=(0, 0, 20), half_extent=6.0)
cam = _capi.make_camera(cd["K"], cd["dist"], cd["R"], cd["t"], size, size)
bvh = engine.BVH(s9)
d_nodes, d_nrm, d_tn = (torch.as_tensor(a).cuda() for a in (verts, nrm, tn))
bvh.set_tri_nodes(d_tn, N)
frames = torch.randint(0, 3000, (F, size, size), device="cuda", dtype=torch.int32).to(torch.uint16)
shard = D.Shard(F, N, 0, 1)
pipe = engine.FramePipeline(1, size, size, N)
K = 4
exch = D.TimeSeriesExchange(shard, K)
bufs = [torch.empty((N, exch.my_chunk(k)[1]), dtype=torch.uint16, device="cuda") for k in range(K)]
T = {}
def tick(name, t0):
    T.setdefault(name, []).append((time.perf_counter() - t0) * 1e6)
first = True
for it in range(6):
    torch.cuda.synchronize()
    ts = time.perf_counter()
    t0 = time.perf_counter(); proj = engine.build_projection(bvh, cam, d_nodes, d_nrm, d_tn, 70.0, counts=False); tick("build_projection", t0)
    t0 = time.perf_counter(); pipe.reset(); tick("reset", t0)
    t0 = time.perf_counter(); pipe.set_projection(0, proj["pix"]); tick("set_projection", t0)
    exch.k = 0
    t0 = time.perf_counter(); sk = engine.skipped_nodes(proj["pix"], want_count=False)[0]; tick("skipped_nodes", t0)
    t0 = time.perf_counter(); exch.set_skipped(sk, assume_same=not first); tick("set_skipped", t0)
    first = False
    t0 = time.perf_counter(); pipe.set_row_map(exch.row_map()); tick("set_row_map", t0)
    nrows = exch.packed_rows()
    for k in range(K):
        c0, fc = exch.my_chunk(k)
        buf = bufs[k][:nrows]
        t0 = time.perf_counter(); pipe.process(frames[c0:c0 + fc], first_frame=c0, rows_t=buf, want_rows=False); tick("process", t0)
        t0 = time.perf_counter(); exch.submit(buf, packed=True); tick("submit", t0)
    t0 = time.perf_counter(); s, ss = pipe.accumulators(); tick("accumulators", t0)
    t0 = time.perf_counter(); exch.finish(); tick("finish", t0)
    t0 = time.perf_counter(); pipe.finalize(F); tick("finalize", t0)
    thost = (time.perf_counter() - ts) * 1e6
    torch.cuda.synchronize()
    print("step %d: host %.0f us, total %.0f us" % (it, thost, (time.perf_counter() - ts) * 1e6))
for k, v in T.items():
    print("%-18s n=%2d  last-step mean %8.1f us   first %8.1f us" % (k, len(v), np.mean(v[len(v) * 5 // 6:]), v[0]))
