"""Debug aid: for a soak seed whose projection ray count differs, cast every retry ray of every
in-frame node through the batch kernel and the oracle and list the rays whose hit differs.
    SOAK_SEED=6438 python tests/debug/dbg_nrays.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from oracle import oracle
from upsp_processing_amd import _capi, engine, synthetic as syn

seed = int(os.environ.get("SOAK_SEED", "6438"))
rng = np.random.default_rng(seed)
assert seed % 5 == 3
g = int(rng.integers(2, 30))
x, y = np.meshgrid(np.arange(g + 1, dtype=np.float32), np.arange(g + 1, dtype=np.float32))
v = np.stack([x.ravel(), y.ravel(), np.zeros(x.size, np.float32)], 1)
q = np.arange(g * g); i0 = q // g * (g + 1) + q % g
t = np.concatenate([np.stack([i0, i0 + 1, i0 + g + 2], 1), np.stack([i0 + g + 2, i0 + g + 1, i0], 1)]).astype(np.int32)
v2 = v.copy(); v2[:, 2] = rng.choice([0.5, 1.0, 3.0])
v = np.concatenate([v, v2]); t = np.concatenate([t, t + x.size]); s9, tn = syn.soup(v, t)
v = np.ascontiguousarray(v, np.float32)
bvh, obv = engine.BVH(s9), oracle.OracleBVH(s9)
m = 4000   # consume the generator exactly like the soak does
org = (rng.normal(size=(m, 3)) * 4).astype(np.float32)
dirs = (-org + rng.normal(size=(m, 3)) * 0.8).astype(np.float32)
tgt = v[rng.integers(0, v.shape[0], m)]
org2 = np.tile((rng.normal(size=(1, 3)) * 6).astype(np.float32), (m, 1))
ax = rng.integers(0, 3, m); dirs3 = np.zeros((m, 3), np.float32); dirs3[np.arange(m), ax] = rng.choice([-1.0, 1.0, 2.5], m)
org3 = (tgt + rng.choice([0.0, 0.0, 0.1], (m, 3)).astype(np.float32))
nrm = syn.node_normals(v, tn.reshape(-1, 3))
W, H = int(rng.choice([64, 200, 512])), int(rng.choice([48, 160, 512]))
cd = syn.pinhole_camera(W, H, center=tuple(rng.normal(size=3) * 3 + np.array([0, 0, 12])), half_extent=float(rng.uniform(2, 7)),
                        k1=float(rng.choice([0.0, -0.05])), azimuth_deg=float(rng.uniform(0, 360)))
cg = _capi.make_camera(cd["K"], cd["dist"], cd["R"], cd["t"], W, H)
co = oracle.make_camera(cd["K"], cd["dist"], cd["R"], cd["t"], W, H)
want = oracle.create_projection(obv, co, v, nrm, tn, engine.oblique_threshold(70.0))
d_tn = torch.as_tensor(np.ascontiguousarray(tn, np.int32)).cuda()
got = engine.build_projection(bvh, cg, v, nrm, d_tn, 70.0)
print("tris", t.shape[0], "nodes", v.shape[0], "image", W, H, "nrays gpu", got["nrays"], "oracle", want["nrays"],
      "primary", got["primary_rays"], "retry nodes", got["retry_nodes"])
cc = np.float32(engine.camera_center(cg))
L = np.float32(1e-4)
sp = np.float32([[-1, 0, 0], [1, 0, 0], [0, -1, 0], [0, 1, 0], [0, 0, -1], [0, 0, 1]])
pos2 = (v[:, None, :] + sp[None] * L).astype(np.float32).reshape(-1, 3)
d2 = (pos2 - cc[None]).astype(np.float32)
o2 = np.tile(cc[None], (d2.shape[0], 1)).astype(np.float32)
a, b = bvh.intersect(o2, d2), obv.intersect(o2, d2)
ah, ap = a["hit"].cpu().numpy(), a["prim"].cpu().numpy()
bad = np.nonzero((ah != b["hit"]) | (ap != b["prim"]))[0]
print("retry rays of all nodes: %d, batch kernel vs oracle differ on %d" % (d2.shape[0], bad.size))
for i in bad[:10]:
    print("  node %d retry %d: gpu hit %d prim %d t %r | oracle hit %d prim %d t %r" %
          (i // 6, i % 6, ah[i], ap[i], a["t"].cpu().numpy()[i], b["hit"][i], b["prim"][i], b["t"][i]))
# primary rays
d1 = (v - cc[None]).astype(np.float32)
d1 = d1 / np.sqrt((d1.astype(np.float32) ** 2).sum(1, dtype=np.float32))[:, None]
pa, pb = engine.project_points(cg, v), oracle.project_points(co, v)
print("project_points host vs oracle equal:", np.array_equal(pa, pb))
r = np.rint(pb).astype(np.int64)
infr = (r[:, 0] >= 0) & (r[:, 1] >= 0) & (r[:, 0] < W) & (r[:, 1] < H)
print("in-frame nodes (oracle arithmetic):", int(infr.sum()))
o1 = np.tile(cc[None], (v.shape[0], 1)).astype(np.float32)
dd = (v - cc[None]).astype(np.float32)
ln = np.sqrt((dd[:, 0] * dd[:, 0] + dd[:, 1] * dd[:, 1]).astype(np.float32) + dd[:, 2] * dd[:, 2]).astype(np.float32)
d1 = (dd / ln[:, None]).astype(np.float32)
a, b = bvh.intersect(o1, d1), obv.intersect(o1, d1)
ah, ap = a["hit"].cpu().numpy(), a["prim"].cpu().numpy()
bad = np.nonzero((ah != b["hit"]) | (ap != b["prim"]))[0]
print("primary rays: batch kernel vs oracle differ on %d" % bad.size)
idx = np.nonzero(infr)[0]
own = lambda prim, n: (tn.reshape(-1, 3)[prim] == n).any()
for n in idx:
    hit, prim = b["hit"][n], b["prim"][n]
    if hit and not own(prim, n):
        print("  oracle: in-frame node %d primary hits foreign prim %d t %r (gpu prim %d t %r); pt %r" %
              (n, prim, b["t"][n], ap[n], a["t"].cpu().numpy()[n], pb[n]))
gp = got["pix"].cpu().numpy()
print("gpu pix>=0:", int((gp >= 0).sum()), "oracle:", int((want["pix"] >= 0).sum()))
