"""Would ordering the primary rays by pixel tile pay?  ("order primary rays by pixel tile (Morton)")
The bench model's NODES renumbered on the host -- mesh order (as built), Morton order of the pixel a node projects to
(8 x 8-pixel tiles, then pixels inside a tile), random order -- so that the dense primary-ray list, which follows the
node numbering, comes out in that order; same geometry, same BVH, same rays.  Kernel times from the library's timers.
A device-side ordering pass would cost ~20 us per build (counting sort, three small launches)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from upsp_processing_amd import _capi, engine, synthetic as syn
size = 1024
verts, tris = syn.tunnel_model() if "uv" in sys.argv[1:] else syn.tunnel_model_quad()
cd = syn.pinhole_camera(size, size, center=(0, 0, 20), half_extent=6.0)
cam = _capi.make_camera(cd["K"], cd["dist"], cd["R"], cd["t"], size, size)
uv = np.asarray(engine.project_points(cam, verts), dtype=np.float64)
u, v = np.clip(np.rint(uv[:, 0]), 0, size - 1).astype(np.int64), np.clip(np.rint(uv[:, 1]), 0, size - 1).astype(np.int64)


def morton(a, b):
    def spread(x):
        x = (x | (x << 8)) & 0x00FF00FF; x = (x | (x << 4)) & 0x0F0F0F0F
        x = (x | (x << 2)) & 0x33333333; x = (x | (x << 1)) & 0x55555555
        return x
    return spread(a) | (spread(b) << 1)


orders = {"mesh order": np.arange(verts.shape[0]),
          "pixel tiles, Morton": np.argsort(morton(u, v), kind="stable"),
          "pixel rows": np.argsort(v * size + u, kind="stable"),
          "random": np.random.default_rng(3).permutation(verts.shape[0])}
for name, perm in orders.items():
    inv = np.empty_like(perm); inv[perm] = np.arange(perm.size)
    vv = np.ascontiguousarray(verts[perm]); tt = np.ascontiguousarray(inv[tris].astype(np.int32))
    s9, tn = syn.soup(vv, tt); nrm = syn.node_normals(vv, tt)
    bvh = engine.BVH(s9)
    d_nodes, d_nrm, d_tn = [torch.as_tensor(x).cuda() for x in (vv, nrm, tn)]
    bvh.set_tri_nodes(d_tn, vv.shape[0])
    for r in range(3):
        engine.build_projection(bvh, cam, d_nodes, d_nrm, d_tn, 70.0, counts=False)
    torch.cuda.synchronize()
    _capi.timing_enable(True)
    for r in range(8):
        engine.build_projection(bvh, cam, d_nodes, d_nrm, d_tn, 70.0, counts=False)
    torch.cuda.synchronize()
    _capi.timing_enable(False)
    rep = _capi.timing_report()
    pc = engine.projection_counts(bvh)
    print("%-20s primary rays %7d retry nodes %6d | " % (name, pc["primary_rays"], pc["retry_nodes"]) +
          "  ".join("%s %.1f us" % (k.replace("projection_kernel", "").replace("_kernel", "").replace("_kernels", ""), v[1] / v[0] * 1e3)
                    for k, v in rep.items() if v[0]), flush=True)
