"""Phase 1 on the reference's own model: test/data/fml_tc3_volume.grid (PLOT3D, 12 zones) seen by
camera01 of the reference's tunnel calibration, synthetic 512 x 1024 frames.  Prints sizes and times."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
import refdata
from upsp_processing_amd import engine, grids, psp, synthetic as syn

G = refdata.GOLDEN
t0 = time.time()
m = grids.P3DModel.from_file(os.path.join(G, "fml_tc3_volume.grid"), 1e-3)
s9, tn = m.extract_tris()
print("model: %d zones, %d nodes, %d triangles, %d overlapping nodes (%.1f s)" %
      (len(m.zones), m.size(), tn.size // 3, len(m.overlap), time.time() - t0))
rmat, tvec, cm, dist = refdata.read_camera_tunnel_cal(os.path.join(G, "camera01_35_6.json"), (512, 1024))
cam = dict(K=cm, dist=np.asarray(dist).ravel()[:4], R=rmat, t=np.asarray(tvec).ravel())
W, H, F = 1024, 512, 256
frames = syn.synth_frames_torch(F, H, W)
t0 = time.time()
job = psp.Phase1(s9, tn, m.nodes(), m.normals, [cam], (W, H), overlap_src=m.overlap_source())
torch.cuda.synchronize()
print("BVH + projection: %.2f s, %d rays, %d of %d nodes seen" %
      (time.time() - t0, job.nrays, int((job.pix[0] >= 0).sum()), m.size()))
for r in range(2):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    p = engine.build_projection(job.bvh, job.cams[0], job.d_nodes, job.d_normals, job.d_tri_nodes, 70.0, counts=False)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
print("projection build alone: %.2f ms" % (dt * 1e3))
job.set_first_frames([frames[0]])
rows_t = torch.empty((m.size(), engine.series_ld(F)), dtype=torch.float32, device="cuda")[:, :F]
for r in range(3):
    job.pipe.reset()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    job.process([frames], first_frame=0, rows_t=rows_t)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
print("frame loop: %.3f ms for %d frames (%.0f frames/s)" % (dt * 1e3, F, F / dt))
fin = job.finalize(F)
avg = fin["avg"]
print("nodes with a finite average: %d ; mean of them: %.1f" % (int(torch.isfinite(avg).sum()), float(avg[torch.isfinite(avg)].mean())))
