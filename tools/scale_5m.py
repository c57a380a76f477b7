import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from upsp_processing_amd import _capi, engine, synthetic as syn
t0 = time.time()
v, t = syn.tunnel_model_quad(576, 205)       # ~5.0 M triangles
s9, tn = syn.soup(v, t); nrm = syn.node_normals(v, t)
print("mesh", v.shape, t.shape, "%.1fs" % (time.time() - t0))
bvh = engine.BVH(s9); print(bvh.info)
size = 1024
dn, dm, dt = [torch.as_tensor(x).cuda() for x in (v, nrm, tn)]
bvh.set_tri_nodes(dt, v.shape[0])
pix = []
for az in (0, 90, 180, 270):
    cd = syn.pinhole_camera(size, size, center=(0, 0, 20), half_extent=6.0, azimuth_deg=az)
    cam = _capi.make_camera(cd["K"], cd["dist"], cd["R"], cd["t"], size, size)
    for r in range(2):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        p = engine.build_projection(bvh, cam, dn, dm, dt, 70.0)
        torch.cuda.synchronize(); dt_ = time.perf_counter() - t0
    for r in range(2):     # the order the frame loops use: oblique test first, no rays for the nodes it rejects
        torch.cuda.synchronize(); t0 = time.perf_counter()
        p2 = engine.build_projection(bvh, cam, dn, dm, dt, 70.0, counts=False)
        torch.cuda.synchronize(); dt2 = time.perf_counter() - t0
    pc = engine.projection_counts(bvh)
    print("cam az=%d: reference order %.2f ms, %d rays (%.0f Mrays/s); oblique test first %.2f ms, %d primary rays + %d retry nodes, "
          "same entries %s; visible %d" % (az, dt_ * 1e3, p["nrays"], p["nrays"] / dt_ / 1e6, dt2 * 1e3, pc["primary_rays"],
                                          pc["retry_nodes"], bool(torch.equal(p["pix"], p2["pix"])), int((p["pix"] >= 0).sum())))
    pix.append(p["pix"])
pix = torch.stack(pix)
centers = np.array([engine.camera_center(_capi.make_camera(*(lambda c: (c["K"], c["dist"], c["R"], c["t"]))(syn.pinhole_camera(size, size, center=(0, 0, 20), half_extent=6.0, azimuth_deg=az)), size, size)) for az in (0, 90, 180, 270)])
w = engine.projection_weights(pix, dn, dm, centers, "average_view")
sk, ns = engine.skipped_nodes(pix)
print("skipped", ns, "of", v.shape[0])
if os.environ.get("UPSP_EXP_PIXEL_ORDER"):
    # experiment: the model's nodes renumbered by the pixel of the first camera that sees them, so that nodes reading
    # the same pixel series are neighbours in pass B
    key = torch.full((v.shape[0],), 1 << 30, dtype=torch.int64, device="cuda")
    for c in range(3, -1, -1):
        key = torch.where(pix[c] >= 0, pix[c].long() + (c << 24), key)
    perm = torch.argsort(key, stable=True)
    pix = pix[:, perm].contiguous(); w = w[:, perm].contiguous(); sk = sk[perm].contiguous()
    print("nodes renumbered in pixel order")
for F in (1000,):
    frames = [syn.synth_frames_torch(F, size, size, first=100 * c) for c in range(4)]
    for mode, name in ((2, "scan + gather"), (1, "streamed (pass A per camera, whole-row pass B)")):
        pipe = engine.FramePipeline(4, size, size, v.shape[0], fused_scan=mode)
        for c in range(4):
            pipe.set_projection(c, pix[c], w[c])
        rows_t = torch.empty((v.shape[0], engine.series_ld(F, whole_rows=(mode == 1))), dtype=torch.float32, device="cuda")[:, :F]
        for r in range(4):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            pipe.process(frames, 0, rows_t=rows_t, want_rows=False)
            torch.cuda.synchronize(); d = time.perf_counter() - t0
        if F == 1000:
            _capi.timing_enable(True)
            pipe.process(frames, 0, rows_t=rows_t, want_rows=False)
            torch.cuda.synchronize()
            _capi.timing_enable(False)
            print("   kernels:", {k: (c, round(ms, 3)) for k, (c, ms) in _capi.timing_report().items()})
        print("4-camera frame loop, %-48s F=%3d: %.2f ms (%.0f camera-frames/s, %.2f TB/s of frames + series)"
              % (name, F, d * 1e3, 4 * F / d, F * (4 * 2 * size * size + 4 * v.shape[0]) / d / 1e12))
        ok = ~sk
        assert torch.isfinite(rows_t[ok]).all().item() and torch.isnan(rows_t[sk]).all().item()
        pipe.close()
    del frames
