"""Interference experiment: projection build on one stream while the frame loop (pass A + pass B of the
previous projection) runs on another.  Prints the serial and the concurrent time of the pair."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from upsp_processing_amd import _capi, engine, synthetic as syn
size, F = 1024, 1000
verts, tris = syn.tunnel_model_quad()
s9, tn = syn.soup(verts, tris); nrm = syn.node_normals(verts, tris); N = verts.shape[0]
cd = syn.pinhole_camera(size, size, center=(0, 0, 20), half_extent=6.0)
cam = _capi.make_camera(cd["K"], cd["dist"], cd["R"], cd["t"], size, size)
bvh = engine.BVH(s9)
d_nodes, d_nrm, d_tn = (torch.as_tensor(a).cuda() for a in (verts, nrm, tn))
bvh.set_tri_nodes(d_tn, N)
frames = torch.randint(0, 3000, (F, size, size), device="cuda", dtype=torch.int32).to(torch.uint16)
pipe = engine.FramePipeline(1, size, size, N)
rt = torch.empty((N, engine.series_ld(F, whole_rows=True)), dtype=torch.float32, device="cuda")[:, :F]
proj = engine.build_projection(bvh, cam, d_nodes, d_nrm, d_tn, 70.0, counts=False)
pipe.set_projection(0, proj["pix"])
pipe.process(frames, 0, rows_t=rt, want_rows=False)
side = torch.cuda.Stream()
main = torch.cuda.current_stream()
def run(mode, n=10):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        if mode == "serial":
            engine.build_projection(bvh, cam, d_nodes, d_nrm, d_tn, 70.0, counts=False)
            pipe.process(frames, 0, rows_t=rt, want_rows=False)
        elif mode == "proj":
            engine.build_projection(bvh, cam, d_nodes, d_nrm, d_tn, 70.0, counts=False)
        elif mode == "frames":
            pipe.process(frames, 0, rows_t=rt, want_rows=False)
        else:
            side.wait_stream(main)
            with torch.cuda.stream(side):
                pipe.process(frames, 0, rows_t=rt, want_rows=False)
            engine.build_projection(bvh, cam, d_nodes, d_nrm, d_tn, 70.0, counts=False)
            main.wait_stream(side)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3
for m in ("serial", "proj", "frames", "concurrent", "serial", "concurrent"):
    run(m, 3)
    print("%-10s %.3f ms" % (m, run(m)))
