#!/usr/bin/env python3
"""Debug: why do jittered retry rays miss the triangle the primary ray hit?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from upsp_processing_amd import _capi, engine, synthetic as syn
v, t = syn.tunnel_model_quad()
s9, tn = syn.soup(v, t)
bvh = engine.BVH(s9)
cam = np.array([0, 0, 20], np.float32)
d = (v - cam).astype(np.float32)
dn = d / np.linalg.norm(d, axis=1, keepdims=True).astype(np.float32)
org = np.broadcast_to(cam, v.shape).copy()
h = bvh.intersect(torch.as_tensor(org).cuda(), torch.as_tensor(dn.astype(np.float32)).cuda(), want=("hit", "t", "prim", "uvw"))
prim = h["prim"].cpu().numpy(); uvw = h["uvw"].cpu().numpy(); tt = h["t"].cpu().numpy()
tri_nodes = tn.reshape(-1, 3)
nid = np.arange(v.shape[0])
own = (tri_nodes[np.maximum(prim, 0)] == nid[:, None]).any(1) & (prim >= 0)
print("nodes", v.shape[0], "primary hits own", own.sum(), "foreign", ((prim >= 0) & ~own).sum())
foreign = np.where((prim >= 0) & ~own)[0]
# retry 0 (x - 1e-4)
q = v[foreign].copy(); q[:, 0] -= 1e-4
d2 = (q - cam).astype(np.float32)
h2 = bvh.intersect(torch.as_tensor(org[foreign]).cuda(), torch.as_tensor(d2).cuda(), want=("hit", "t", "prim", "uvw"))
p2 = h2["prim"].cpu().numpy()
same = p2 == prim[foreign]
print("retry0 hits the same triangle as the primary: %.1f %%" % (100 * same.mean()))
miss = foreign[~same]
print("min barycentric of the primary hit on W, nodes whose retry misses W: median %.2e" % np.median(uvw[miss].min(1)),
      " (others: %.2e)" % np.median(uvw[foreign[same]].min(1)))
dist = np.linalg.norm(v[miss] - (cam + dn[miss] * tt[miss, None]), axis=1)
print("distance primary hit -> node, missing: median %.3g  p10 %.3g p90 %.3g" % (np.median(dist), np.percentile(dist, 10), np.percentile(dist, 90)))
dist2 = np.linalg.norm(v[foreign[same]] - (cam + dn[foreign[same]] * tt[foreign[same], None]), axis=1)
print("distance primary hit -> node, same: median %.3g" % np.median(dist2))
print("z of missing nodes: median %.3f ; z of same: %.3f" % (np.median(v[miss, 2]), np.median(v[foreign[same], 2])))
# how are the triangle the retry hits and W related?
a = tri_nodes[prim[miss]]; b = tri_nodes[p2[~same]]
valid = p2[~same] >= 0
shared = np.zeros(len(miss), int)
for i in range(3):
    for j in range(3):
        shared += (a[:, i] == b[:, j])
print("retry0 misses W: closest hit shares 2 nodes with W: %.1f %%, 1 node: %.1f %%, 0: %.1f %%, no hit %.1f %%" % (
    100 * (shared[valid] == 2).mean(), 100 * (shared[valid] == 1).mean(), 100 * (shared[valid] == 0).mean(), 100 * (~valid).mean()))
# which edge: is the shared edge the one opposite the smallest barycentric of the primary hit?
