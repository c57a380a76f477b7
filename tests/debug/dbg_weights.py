import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from oracle import oracle as orc
from upsp_processing_amd import synthetic as syn, engine, _capi
v, t = syn.tunnel_model(60, 120, 24, 48)
s9, tn = syn.soup(v, t); nrm = syn.node_normals(v, t)
obv = orc.OracleBVH(s9)
po=[]; centers=[]
for az in (0, 60, 120, 200):
    c = syn.pinhole_camera(512, 512, center=(0.37 * az / 60, 0.2, 20 + az / 50), half_extent=6.0, azimuth_deg=az)
    co = orc.make_camera(c["K"], c["dist"], c["R"], c["t"], 512, 512)
    po.append(orc.create_projection(obv, co, v, nrm, tn, engine.oblique_threshold(70.0))["pix"])
    centers.append(orc.cam_center(co))
po=np.stack(po); centers=np.array(centers)
for mode, m in (("best_view", 0), ("average_view", 1)):
    wg = engine.projection_weights(torch.as_tensor(po).cuda(), v, nrm, centers, mode).cpu().numpy()
    wo = orc.adjust_weights(po, np.ones_like(po, dtype=np.float32), v, nrm, centers, m)
    bad = ~np.isclose(wg, wo, rtol=2e-7, atol=0)
    print(mode, "bad entries", bad.sum(), "nodes", bad.any(0).sum())
    i = np.nonzero(bad.any(0))[0][:4]
    print(i, "\nwg", wg[:, i], "\nwo", wo[:, i], "\npix", po[:, i])
