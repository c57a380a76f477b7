#!/usr/bin/env python3
"""Round 6: A/B of the traversal changes in ONE process (GPU).  The slab filter is read per launch (UPSP_SLAB_FILTER), so it is
switched through os.environ between timed blocks; the ray bins are read once per process (UPSP_RAY_BINS in the environment).

  python tools/r06_rays.py            -> 1 Mi pixel rays on the frame-filling sphere and the tunnel model's projection build, slab on / off
"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench  # noqa: E402
from upsp_processing_amd import _capi, engine, synthetic as syn  # noqa: E402


def timed(fn, reps):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def main():
    size = 1024
    fv, ft = syn.cube_sphere(289, 6.0)
    fs9, _ = syn.soup(fv, ft)
    fcd = syn.pinhole_camera(size, size, center=(0, 0, 20), half_extent=6.0, fill=0.95)
    fbvh = engine.BVH(fs9)
    org, dirs = bench.pixel_rays(fcd, size)
    d_org, d_dirs = torch.as_tensor(org).cuda(), torch.as_tensor(dirs).cuda()
    verts, tris = syn.tunnel_model_quad()
    s9, tn = syn.soup(verts, tris)
    nrm = syn.node_normals(verts, tris)
    cd = syn.pinhole_camera(size, size, center=(0, 0, 20), half_extent=6.0, fill=0.7)
    cam = _capi.make_camera(cd["K"], cd["dist"], cd["R"], cd["t"], size, size)
    bvh = engine.BVH(s9)
    d_nodes, d_nrm, d_tn = torch.as_tensor(verts).cuda(), torch.as_tensor(nrm).cuda(), torch.as_tensor(tn).cuda()
    bvh.set_tri_nodes(d_tn, verts.shape[0])
    ref = {}
    for rep in range(3):
        for slab in ("1", "0"):
            os.environ["UPSP_SLAB_FILTER"] = slab
            ms_fill = timed(lambda: fbvh.intersect(d_org, d_dirs, want=("hit", "t", "prim")), 10)
            ms_build = timed(lambda: engine.build_projection(bvh, cam, d_nodes, d_nrm, d_tn, 70.0, counts=False), 10)
            h = fbvh.intersect(d_org, d_dirs, want=("hit", "t", "prim"))
            key = (h["t"].view(torch.int32).sum().item(), h["prim"].sum().item(), h["hit"].sum().item())
            pix = engine.build_projection(bvh, cam, d_nodes, d_nrm, d_tn, 70.0, counts=False)["pix"]
            key = key + (int(pix.sum().item()),)
            ref.setdefault("key", key)
            print("rep %d slab %s: 1 Mi pixel rays (fill) %.4f ms = %.0f Mrays/s; projection build alone %.4f ms; same results %s" % (
                rep, slab, ms_fill, dirs.shape[0] / ms_fill / 1e3, ms_build, key == ref["key"]), flush=True)
    os.environ["UPSP_SLAB_FILTER"] = "1"
    fbvh.enable_stats(True)
    fbvh.intersect(d_org, d_dirs, want=("hit",))
    st, fs = fbvh.last_stats(), fbvh.last_filter_stats()
    print("fill scene: %.2f wide steps per ray; slab filter saw %d boxes, left %d undecided (%.4f %%)" % (
        st["nodes"] / dirs.shape[0], fs["boxes"], fs["undecided"], 100.0 * fs["undecided"] / max(fs["boxes"], 1)), flush=True)
    _capi.timing_enable(True)
    for _ in range(5):
        engine.build_projection(bvh, cam, d_nodes, d_nrm, d_tn, 70.0, counts=False)
    torch.cuda.synchronize()
    rep = _capi.timing_report()
    print({k: round(v[1] / max(v[0], 1), 4) for k, v in rep.items()}, flush=True)


if __name__ == "__main__":
    main()
