"""Randomised parity soak (run on a GPU box from the repository root):
    python tests/debug/soak_raycast.py [seconds]
Random scenes (soups, spheres, tunnel models, flat / degenerate pieces), random and adversarial
rays (aimed at vertices, axis-aligned, zero components, origins inside boxes), random cameras
for the projection build with and without the node->triangle adjacency -- GPU vs oracle, bit-exact.
Prints one line per scene; exits non-zero on the first mismatch."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from oracle import oracle
from upsp_processing_amd import _capi, engine, synthetic as syn

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
t_end = time.time() + budget
seed = int(os.environ.get("SOAK_SEED", "1000"))
nscenes = nrays_total = 0


def same(a, b):
    a = a.cpu().numpy() if hasattr(a, "cpu") else a
    return (a.view(np.int32) == b.view(np.int32)).all() if a.dtype == np.float32 else (a == b).all()


while time.time() < t_end:
    rng = np.random.default_rng(seed)
    kind = seed % 5
    if kind == 0:
        n = int(rng.integers(1, 4000))
        c = rng.normal(size=(n, 1, 3)) * 3
        s9 = (c + rng.normal(size=(n, 3, 3)) * rng.choice([0.05, 0.5, 2.0])).astype(np.float32).reshape(-1)
        v = s9.reshape(-1, 3); tn = np.arange(v.shape[0], dtype=np.int32)
    elif kind == 1:
        v, t = syn.uv_sphere(int(rng.integers(2, 40)), int(rng.integers(3, 80))); s9, tn = syn.soup(v, t)
    elif kind == 2:
        v, t = syn.tunnel_model_quad(int(rng.integers(4, 40)), int(rng.integers(2, 16))); s9, tn = syn.soup(v, t)
    elif kind == 3:   # axis-aligned plates: flat boxes, many exact ties
        g = int(rng.integers(2, 30))
        x, y = np.meshgrid(np.arange(g + 1, dtype=np.float32), np.arange(g + 1, dtype=np.float32))
        v = np.stack([x.ravel(), y.ravel(), np.zeros(x.size, np.float32)], 1)
        q = np.arange(g * g); i0 = q // g * (g + 1) + q % g
        t = np.concatenate([np.stack([i0, i0 + 1, i0 + g + 2], 1), np.stack([i0 + g + 2, i0 + g + 1, i0], 1)]).astype(np.int32)
        v2 = v.copy(); v2[:, 2] = rng.choice([0.5, 1.0, 3.0])
        v = np.concatenate([v, v2]); t = np.concatenate([t, t + x.size]); s9, tn = syn.soup(v, t)
    else:
        v, t = syn.cube_sphere(int(rng.integers(2, 30)), 1.0, scale=(rng.uniform(1, 6), 1, 1)); s9, tn = syn.soup(v, t)
    v = np.ascontiguousarray(v, np.float32)
    bvh, obv = engine.BVH(s9), oracle.OracleBVH(s9)
    ok = bvh.info["n_ref_nodes"] == obv.nnodes and bvh.info["depth"] == obv.depth
    # rays: random, at vertices, axis-aligned with zero components
    m = 4000
    org = (rng.normal(size=(m, 3)) * 4).astype(np.float32)
    dirs = (-org + rng.normal(size=(m, 3)) * 0.8).astype(np.float32)
    tgt = v[rng.integers(0, v.shape[0], m)]
    org2 = np.tile((rng.normal(size=(1, 3)) * 6).astype(np.float32), (m, 1)); dirs2 = (tgt - org2).astype(np.float32)
    ax = rng.integers(0, 3, m); dirs3 = np.zeros((m, 3), np.float32); dirs3[np.arange(m), ax] = rng.choice([-1.0, 1.0, 2.5], m)
    org3 = (tgt + rng.choice([0.0, 0.0, 0.1], (m, 3)).astype(np.float32)); org3[np.arange(m), ax] -= dirs3[np.arange(m), ax] * 3
    for o_, d_ in ((org, dirs), (org2, dirs2), (org3.astype(np.float32), dirs3)):
        g, o = bvh.intersect(o_, d_), obv.intersect(o_, d_)
        ok = ok and np.array_equal(g["hit"].cpu().numpy(), o["hit"]) and all(same(g[k], o[k]) for k in ("t", "prim", "uvw", "pos", "nrm"))
        nrays_total += m
    # projection build with / without adjacency
    if tn.size == 3 * (s9.size // 9) and v.shape[0] >= 3:
        nrm = syn.node_normals(v, tn.reshape(-1, 3)) if kind != 0 else np.tile(np.float32([0, 0, 1]), (v.shape[0], 1))
        W, H = int(rng.choice([64, 200, 512])), int(rng.choice([48, 160, 512]))
        cd = syn.pinhole_camera(W, H, center=tuple(rng.normal(size=3) * 3 + np.array([0, 0, 12])), half_extent=float(rng.uniform(2, 7)),
                                k1=float(rng.choice([0.0, -0.05])), azimuth_deg=float(rng.uniform(0, 360)))
        cg = _capi.make_camera(cd["K"], cd["dist"], cd["R"], cd["t"], W, H)
        co = oracle.make_camera(cd["K"], cd["dist"], cd["R"], cd["t"], W, H)
        want = oracle.create_projection(obv, co, v, nrm, tn, engine.oblique_threshold(70.0))
        d_tn = torch.as_tensor(np.ascontiguousarray(tn, np.int32)).cuda()
        for adj in (False, True):
            if adj:
                bvh.set_tri_nodes(d_tn, v.shape[0])
            g = engine.build_projection(bvh, cg, v, nrm, d_tn, 70.0)
            okp = np.array_equal(g["pix"].cpu().numpy(), want["pix"]) and g["nrays"] == want["nrays"] \
                and same(g["uv"], want["uv"])
            if not okp:
                gp = g["pix"].cpu().numpy()
                bad = np.nonzero(gp != want["pix"])[0]
                print("  projection mismatch (adjacency %s): %d nodes differ, first %s gpu %s oracle %s; nrays %d vs %d"
                      % (adj, bad.size, bad[:5], gp[bad[:5]], want["pix"][bad[:5]], g["nrays"], want["nrays"]), flush=True)
            ok = ok and okp
            # the order the frame loops use: oblique test first, no rays for the nodes it rejects
            g = engine.build_projection(bvh, cg, v, nrm, d_tn, 70.0, counts=False)
            okc = np.array_equal(g["pix"].cpu().numpy(), want["pix"]) and same(g["uv"], want["uv"])
            if not okc:
                print("  projection mismatch with the oblique test first (adjacency %s)" % adj, flush=True)
            ok = ok and okc
    nscenes += 1
    print("seed %d kind %d tris %d: %s" % (seed, kind, s9.size // 9, "ok" if ok else "MISMATCH"), flush=True)
    bvh.close()
    if not ok:
        sys.exit(1)
    seed += 1
print("soak: %d scenes, %d rays, all bit-exact" % (nscenes, nrays_total))
