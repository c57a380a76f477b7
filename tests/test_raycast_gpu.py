"""GPU parity: HIP ray caster (through the C ABI and the pybind11 `raycast` module)
vs the CPU oracle on identical inputs.  Bar: bit-exact hit flag, t, primID, u/v/w,
pos, nrm (integer / IEEE +-*/ work; see DESIGN.md 'parity')."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def rand_rays(n, seed, spread=3.0, jitter=0.6):
    rng = np.random.default_rng(seed)
    org = (rng.normal(size=(n, 3)) * spread).astype(np.float32)
    dirs = (-org + rng.normal(size=(n, 3)) * jitter).astype(np.float32)
    return org, dirs


def assert_hits_equal(g, o):
    hit_g = g["hit"].cpu().numpy() if hasattr(g["hit"], "cpu") else g["hit"]
    assert np.array_equal(hit_g, o["hit"])
    for k in ("t", "prim", "uvw", "pos", "nrm"):
        a = g[k].cpu().numpy() if hasattr(g[k], "cpu") else g[k]
        b = o[k]
        same = (a.view(np.int32) == b.view(np.int32)) if a.dtype == np.float32 else (a == b)
        assert same.all(), (k, np.argwhere(~same)[:5], a[~same][:5], b[~same][:5])


@pytest.mark.parametrize("nlat,nlon", [(2, 3), (12, 24), (50, 100)])
def test_closest_hit_sphere_bitwise(gpu_lib, oracle, nlat, nlon):
    from upsp_processing_amd import engine, synthetic as syn
    v, t = syn.uv_sphere(nlat, nlon)
    s9, _ = syn.soup(v, t)
    bvh = engine.BVH(s9)
    obv = oracle.OracleBVH(s9)
    assert bvh.info["n_ref_nodes"] == obv.nnodes and bvh.info["depth"] == obv.depth
    org, dirs = rand_rays(20000, nlat)
    assert_hits_equal(bvh.intersect(org, dirs), obv.intersect(org, dirs))
    # shared origin (camera rays), un-normalised directions
    o1 = np.array([0.1, -0.2, 4.0], np.float32)
    d2 = (np.random.default_rng(1).normal(size=(5000, 3)) * [0.3, 0.3, 0.05] + [0, 0, -1]).astype(np.float32) * 7.5
    assert_hits_equal(bvh.intersect(o1, d2), obv.intersect(o1, d2))
    occ = bvh.occluded(org, dirs).cpu().numpy()
    assert np.array_equal(occ, obv.intersect(org, dirs)["hit"])


def test_tunnel_model_and_vertex_rays_bitwise(gpu_lib, oracle):
    """Rays aimed exactly at mesh vertices (the create_projection_mat pattern: grazing
    box corners, exact-zero edge functions -> double fallback, equal-t ties)."""
    from upsp_processing_amd import engine, synthetic as syn
    v, t = syn.tunnel_model(60, 120, 24, 48)
    s9, _ = syn.soup(v, t)
    bvh = engine.BVH(s9)
    obv = oracle.OracleBVH(s9)
    cam = np.array([0.5, 3.0, 20.0], np.float32)
    d = (v - cam).astype(np.float32)
    assert_hits_equal(bvh.intersect(cam, d), obv.intersect(cam, d))
    dn = d / np.linalg.norm(d, axis=1, keepdims=True).astype(np.float32)
    assert_hits_equal(bvh.intersect(cam, dn), obv.intersect(cam, dn))
    # axis-aligned rays: zero direction components, origins inside / on boxes
    org = np.repeat(v[::7], 3, axis=0).astype(np.float32)
    dirs = np.tile(np.eye(3, dtype=np.float32), (org.shape[0] // 3, 1))
    assert_hits_equal(bvh.intersect(org, dirs), obv.intersect(org, dirs))
    assert_hits_equal(bvh.intersect(org, -dirs), obv.intersect(org, -dirs))


def test_degenerate_scenes(gpu_lib, oracle):
    from upsp_processing_amd import engine, _capi
    # single triangle (root is a leaf), coplanar axis-aligned quad pair (flat boxes)
    tri = np.array([0, 0, 1, 0, 1, 0, 1, 0, 0], np.float32)
    quad = np.array([0, 0, 2, 1, 0, 2, 1, 1, 2, 0, 0, 2, 1, 1, 2, 0, 1, 2], np.float32)
    # 200 identical triangles: zero centroid extent -> one oversized leaf (pspRT.cpp:485-494)
    many = np.tile(tri, 200)
    for s9 in (tri, quad, many, np.concatenate([tri, quad, many])):
        bvh, obv = engine.BVH(s9), oracle.OracleBVH(s9)
        org, dirs = rand_rays(4000, 9, spread=2.0, jitter=1.0)
        assert_hits_equal(bvh.intersect(org, dirs), obv.intersect(org, dirs))
    with pytest.raises(_capi.UpspError):
        engine.BVH(np.zeros(0, np.float32))        # reference: "no primitives!" then DIE
    bvh = engine.BVH(tri)
    out = bvh.intersect(np.zeros((0, 3), np.float32), np.zeros((0, 3), np.float32))
    assert out["t"].numel() == 0


def test_reference_toy_kats_through_pybind(gpu_lib):
    """test/python/test_visibility.py toy scenes through the drop-in `raycast` module."""
    from upsp_processing_amd import raycast
    from upsp_processing_amd.visibility import VisibilityChecker
    t0 = [0., 0., 1., 0., 1., 0., 1., 0., 0.]
    t1 = [0., 0., 1., 0., 1., 0., 0., 1., 1.]
    scene = raycast.CreateBVH(t0 + t1, 3)
    assert type(scene).__name__ == "BVH"       # external_calibrate.py:1453
    vc = VisibilityChecker(scene, oblique_angle=70, epsilon=1e-4)
    A = np.array
    cam = A([1., 1., 1.]).reshape(3, 1)
    assert vc.is_visible(cam, A([[2., 2, 2], [3, 3, 3]]), A([[1., 1, 1], [1, 1, 1]])).tolist() == []
    assert vc.is_visible(cam, A([[0., 0, 0], [-1, -1, -1]]), A([[.9, .9, .9], [1.1, 1.1, 1.1]])).tolist() == []
    assert vc.is_visible(A([-8., -8, 0]), A([[-5., -5, 0], [-5, -1, 0]]), A([[-.9, 0, 0], [0, -1.2, 0]])).tolist() == [0, 1]
    assert vc.is_visible(A([-2., -2, -2]).reshape(3, 1), A([[-1., -1, -1], [0, 0, 0]]), -np.ones((2, 3))).tolist() == [0, 1]
    nodes = A([[2., 2, 2], [3, 3, 3], [.9, .9, .9], [.5, .5, .5], [0, 0, 0], [-1, -1, -1]])
    assert vc.is_visible(cam, nodes, np.ones((6, 3))).tolist() == [2, 3]
    # single-ray API exactly as visibility.py:410-412 uses it
    r = raycast.Ray(0.25, 0.25, 5.0, 0.0, 0.0, -1.0)
    h = raycast.Hit()
    assert scene.intersect(r, h) is True
    assert np.allclose(h.pos, [0.25, 0.25, 0.5])
    h2 = raycast.Hit()
    assert scene.intersect(raycast.Ray(5, 5, 5, 0, 0, 1), h2) is False
    assert h2.pos == [0.0, 0.0, 0.0]


def test_camera01_regression_gpu(gpu_lib, fml):
    """test_camera01 (test/python/test_visibility.py:243-254): 148 608 visible nodes on the
    reference's 609 120-triangle grid, and the same index set as the oracle."""
    import refdata
    from upsp_processing_amd import raycast
    from upsp_processing_amd.visibility import VisibilityChecker
    scene = raycast.CreateBVH(fml["prims"], 3)
    vc = VisibilityChecker(scene, oblique_angle=70, epsilon=1e-4)
    vis = vc.is_visible(fml["cam_t"], fml["nodes"], fml["norms"])
    assert len(vis) == 148608
    golden = np.load(os.path.join(refdata.GOLDEN, "camera01_visible.npz"))["visible"]
    assert np.array_equal(vis, golden)


def test_photogrammetry_hit_position(gpu_lib, fml, oracle):
    """Closest-hit positions on the reference grid: GPU == oracle bit for bit on 50k rays
    from camera01 towards grid nodes (the does_intersect(return_pos=True) path,
    target_bumping.py:54)."""
    from upsp_processing_amd import engine
    bvh = engine.BVH(fml["prims"])
    obv = oracle.OracleBVH(fml["prims"])
    cam = fml["cam_t"].reshape(3).astype(np.float32)
    d = (fml["nodes"][::6] - cam).astype(np.float32)
    assert_hits_equal(bvh.intersect(cam, d), obv.intersect(cam, d))


def test_large_batch_two_pass_bitwise(gpu_lib, oracle):
    """Batches of >= 65 536 rays go through cast_entry_kernel first (rays that miss the root box get their record
    there, the rest a dense list with a static assignment).  Pixel-style rays of which most miss the model, rays from
    inside the root box, per-ray origins -- every record against the oracle; occluded() alike."""
    import os
    from upsp_processing_amd import engine, synthetic as syn
    v, t = syn.tunnel_model(60, 120, 24, 48)
    s9, _ = syn.soup(v, t)
    bvh = engine.BVH(s9)
    obv = oracle.OracleBVH(s9)
    thr = os.cpu_count() or 1
    rng = np.random.default_rng(5)
    n = 150_000
    cam = np.array([0.3, -0.4, 20.0], np.float32)
    tgt = (rng.random((n, 3)) - 0.5) * np.array([40.0, 30.0, 2.0])        # most pass beside the model
    d = (tgt - cam).astype(np.float32)
    g, o = bvh.intersect(cam, d), obv.intersect(cam, d, threads=thr)
    assert 0.02 < o["hit"].mean() < 0.6
    assert_hits_equal(g, o)
    assert np.array_equal(bvh.occluded(cam, d).cpu().numpy(), o["hit"])
    org = (rng.normal(size=(n, 3)) * np.array([8.0, 3.0, 3.0])).astype(np.float32)   # many origins inside the root box
    d2 = rng.normal(size=(n, 3)).astype(np.float32)
    d2[::11, 0] = 0.0
    g, o = bvh.intersect(org, d2), obv.intersect(org, d2, threads=thr)
    assert_hits_equal(g, o)
    assert np.array_equal(bvh.occluded(org, d2).cpu().numpy(), o["hit"])


@pytest.mark.parametrize("steps", [3, 40])
def test_heavy_rays_of_a_batch(gpu_lib, oracle, monkeypatch, steps):
    """Batch queries hand rays that exceed UPSP_HEAVY_STEPS_CAST node visits + triangle tests (default 256: rays through
    the 1000-triangle polar fans of a UV sphere) to a kernel in which a whole wave walks one ray.  With a threshold of a
    few steps most rays take that road: full hit records and occlusion flags against the oracle, small and large batches
    (the latter through the two-pass form), rays aimed at the poles."""
    import os
    from upsp_processing_amd import engine, synthetic as syn
    monkeypatch.setenv("UPSP_HEAVY_STEPS_CAST", str(steps))
    v, t = syn.tunnel_model(60, 300, 24, 48)
    s9, _ = syn.soup(v, t)
    bvh, obv = engine.BVH(s9), oracle.OracleBVH(s9)
    thr = os.cpu_count() or 1
    rng = np.random.default_rng(11)
    cam = np.array([0.2, 0.1, 20.0], np.float32)
    for n in (6000, 90_000):
        tgt = v[rng.integers(0, v.shape[0], n)] + rng.normal(size=(n, 3)).astype(np.float32) * rng.choice([0.0, 1e-3, 0.4], (n, 1))
        tgt[::7] = v[np.argmax(np.abs(v[:, 0]))]            # a pole of the stretched sphere
        d = (tgt - cam).astype(np.float32)
        g, o = bvh.intersect(cam, d), obv.intersect(cam, d, threads=thr)
        assert_hits_equal(g, o)
        assert np.array_equal(bvh.occluded(cam, d).cpu().numpy(), o["hit"])


def test_walks_end_with_an_error_not_a_hang(gpu_lib, oracle, monkeypatch):
    """Every walk carries a round cap (2 x nodes on a well-formed tree, which no ray can reach); a walk that runs past
    it sets an error flag and ENDS, and the next synchronising entry point returns UPSP_ERR_INTERNAL -- the reference
    DIEs on a broken BVH (pspRT.cpp:362-365), the device is never wedged.  Forced here with UPSP_ROUND_CAP = 2 rounds:
    the one-lane traversal (batch query, projection build), the cooperative walk (every ray handed over) and the
    explicit check; afterwards, with the cap back at its real value, the same calls are bit-exact again."""
    import torch
    from upsp_processing_amd import _capi, engine, synthetic as syn
    v, t = syn.uv_sphere(40, 80)
    s9, tn = syn.soup(v, t)
    nrm = syn.node_normals(v, t)
    bvh, obv = engine.BVH(s9), oracle.OracleBVH(s9)
    cam = np.array([0.1, 0.2, 20.0], np.float32)
    d = (v[::3] - cam).astype(np.float32)
    c = syn.pinhole_camera(256, 256)
    cam_g = _capi.make_camera(c["K"], c["dist"], c["R"], c["t"], 256, 256)
    d_nodes, d_nrm, d_tn = [torch.as_tensor(a).cuda() for a in (v, nrm, tn)]

    def expect_internal(fn):
        with pytest.raises(_capi.UpspError) as e:
            fn()
        assert e.value.status == -7, e.value

    monkeypatch.setenv("UPSP_ROUND_CAP", "2")
    bvh.intersect(torch.as_tensor(cam).cuda(), torch.as_tensor(d).cuda())      # device buffers: no sync, no error yet ...
    expect_internal(bvh.check)                                                  # ... reported at the caller's sync point
    bvh.check()                                                                 # (reported once)

    def host_query():                                                           # host buffers: the call synchronises itself
        import ctypes as C
        n = d.shape[0]
        tt = np.zeros(n, np.float32)
        h = _capi.Hits()
        h.t = tt.ctypes.data
        _capi.check(gpu_lib.upsp_bvh_intersect_host(bvh.handle, cam.ctypes.data_as(C.c_void_p), 0,
                                                    d.ctypes.data_as(C.c_void_p), n, C.byref(h)))
    expect_internal(host_query)
    expect_internal(lambda: engine.build_projection(bvh, cam_g, d_nodes, d_nrm, d_tn, 70.0, counts=True))
    monkeypatch.setenv("UPSP_HEAVY_STEPS_CAST", "1")                           # every ray to the cooperative walk
    bvh.intersect(torch.as_tensor(cam).cuda(), torch.as_tensor(d).cuda())
    expect_internal(bvh.check)
    monkeypatch.delenv("UPSP_ROUND_CAP")
    g, o = bvh.intersect(cam, d), obv.intersect(cam, d)
    assert_hits_equal(g, o)
    bvh.check()
    monkeypatch.delenv("UPSP_HEAVY_STEPS_CAST")
    assert_hits_equal(bvh.intersect(cam, d), o)
    p = engine.build_projection(bvh, cam_g, d_nodes, d_nrm, d_tn, 70.0, counts=True)
    cam_o = oracle.make_camera(c["K"], c["dist"], c["R"], c["t"], 256, 256)
    po = oracle.create_projection(obv, cam_o, v, nrm, tn, engine.oblique_threshold(70.0))
    assert np.array_equal(p["pix"].cpu().numpy(), po["pix"]) and p["nrays"] == po["nrays"]
    bvh.check()


def test_full_size_properties(gpu_lib):
    """BASELINE config size (1 M-tri model, 1 Mi rays): size-independent properties.
    * scale invariance of the hit set: d and 2d hit the same triangle with t/2;
    * every reported hit re-verifies: pos = o + t d lies in the triangle's plane;
    * occluded() == intersect().hit."""
    import torch
    from upsp_processing_amd import engine, synthetic as syn
    v, t = syn.tunnel_model()
    s9, _ = syn.soup(v, t)
    bvh = engine.BVH(s9)
    assert bvh.info["ntris"] == 1001520
    n = 1 << 20
    g = torch.Generator(device="cuda"); g.manual_seed(7)
    cam = torch.tensor([0.0, 0.0, 20.0], device="cuda")
    tgt = (torch.rand((n, 3), generator=g, device="cuda") - 0.5) * torch.tensor([14.0, 4.0, 2.0], device="cuda")
    d = tgt - cam
    a = bvh.intersect(cam, d)
    b = bvh.intersect(cam, 2.0 * d)
    assert torch.equal(a["hit"], b["hit"]) and torch.equal(a["prim"], b["prim"])
    h = a["hit"]
    assert 0.2 < h.float().mean().item() < 0.9
    assert torch.allclose(a["t"][h], 2.0 * b["t"][h], rtol=1e-6)
    assert torch.equal(bvh.occluded(cam, d), h)
    tri = torch.as_tensor(s9, device="cuda").reshape(-1, 3, 3)[a["prim"][h].long()]
    nrm = torch.cross(tri[:, 1] - tri[:, 0], tri[:, 2] - tri[:, 0], dim=1)
    nrm = nrm / nrm.norm(dim=1, keepdim=True)
    dist = ((a["pos"][h] - tri[:, 0]) * nrm).sum(1).abs()
    assert dist.max().item() < 1e-4
    uvw = a["uvw"][h]
    assert (uvw.sum(1) - 1).abs().max().item() < 1e-5 and uvw.min().item() >= 0


@pytest.mark.parametrize("seed", [0, 1, 2, 3])
def test_random_soups_bitwise(gpu_lib, oracle, seed):
    """Unstructured triangle soups (overlapping, sliver, zero-area, duplicated, axis-aligned
    triangles; coordinates on a coarse lattice so that exact ties and exact-zero edge functions
    are common) x rays with zero / negative-zero / huge / tiny components."""
    from upsp_processing_amd import engine
    rng = np.random.default_rng(seed)
    ntri = [7, 300, 5000, 20000][seed]
    lattice = rng.integers(-8, 9, size=(ntri, 3, 3)).astype(np.float32) * 0.25
    smooth = rng.normal(size=(ntri, 3, 3)).astype(np.float32)
    tris = np.where(rng.random((ntri, 1, 1)) < 0.5, lattice, smooth * 2)
    tris[::11, 2] = tris[::11, 1]                       # zero-area (two equal vertices)
    tris[::13] = tris[1::13][: len(tris[::13])] if len(tris[1::13]) >= len(tris[::13]) else tris[::13]  # duplicates
    tris[::17, :, 2] = 0.5                              # axis-aligned (flat boxes)
    s9 = np.ascontiguousarray(tris.reshape(-1), np.float32)
    bvh, obv = engine.BVH(s9), oracle.OracleBVH(s9)
    assert bvh.info["n_ref_nodes"] == obv.nnodes
    n = 30000
    org = (rng.integers(-12, 13, size=(n, 3)) * 0.25).astype(np.float32)
    dirs = rng.normal(size=(n, 3)).astype(np.float32)
    dirs[::5] = rng.integers(-2, 3, size=(len(dirs[::5]), 3)).astype(np.float32)   # lattice directions, zeros
    dirs[1::7, 0] = -0.0
    dirs[2::9] *= 1e6
    dirs[3::9] *= 1e-6
    dirs[(dirs == 0).all(1)] = [0, 0, 1]
    # rays aimed exactly at vertices
    tgt = tris.reshape(-1, 3)[rng.integers(0, ntri * 3, size=n // 3)]
    dirs[: n // 3] = tgt - org[: n // 3]
    dirs[(dirs == 0).all(1)] = [1, 0, 0]
    assert_hits_equal(bvh.intersect(org, dirs), obv.intersect(org, dirs))
    assert np.array_equal(bvh.occluded(org, dirs).cpu().numpy(), obv.intersect(org, dirs)["hit"])


def test_slab_filter_same_bits(gpu_lib, oracle, monkeypatch):
    """Round 6: the one-lane traversal decides most boxes with a plain slab test and a per-ray error bound (box_filter_slab) and
    hands the rest to the mirrored reciprocal filter / Imath's divisions.  Closest hits (every field) on vertex-aimed rays, random
    rays and a lattice of rays that graze box faces must be the oracle's with the filter on, off (UPSP_SLAB_FILTER=0) and with its
    band widened 10^5 times (UPSP_SLAB_SCALE: nearly every box goes through the fallback); the statistics report how many boxes
    the filter left undecided: a small share by default, most of them with the wide band."""
    from upsp_processing_amd import engine, synthetic as syn
    v, t = syn.tunnel_model(40, 80, 16, 32)
    s9, _ = syn.soup(v, t)
    obv = oracle.OracleBVH(s9)
    cam = np.array([0.5, 3.0, 20.0], np.float32)
    d_vertex = (v - cam).astype(np.float32)
    org, dirs = rand_rays(20000, 5, spread=8.0, jitter=1.5)
    # rays through box corners / along faces: origins and targets on the vertex lattice of the model
    rng = np.random.default_rng(11)
    a, b = v[rng.integers(0, v.shape[0], 8000)], v[rng.integers(0, v.shape[0], 8000)]
    org_l, dir_l = (a + (a - b) * 0.5).astype(np.float32), (b - a).astype(np.float32)
    want = [obv.intersect(cam, d_vertex), obv.intersect(org, dirs), obv.intersect(org_l, dir_l)]
    shares = {}
    for name, env in (("default", {}), ("off", {"UPSP_SLAB_FILTER": "0"}), ("wide", {"UPSP_SLAB_SCALE": "100000"})):
        for k in ("UPSP_SLAB_FILTER", "UPSP_SLAB_SCALE"):
            monkeypatch.delenv(k, raising=False)
        for k, val in env.items():
            monkeypatch.setenv(k, val)
        bvh = engine.BVH(s9)
        bvh.enable_stats(True)
        for (o, d), w in zip(((cam, d_vertex), (org, dirs), (org_l, dir_l)), want):
            assert_hits_equal(bvh.intersect(o, d), w)
        bvh.intersect(org, dirs)
        shares[name] = bvh.last_filter_stats()
        bvh.close()
    assert shares["off"]["boxes"] == 0
    assert shares["default"]["boxes"] > 0 and shares["default"]["undecided"] < 0.02 * shares["default"]["boxes"], shares
    assert shares["wide"]["undecided"] > 0.3 * shares["wide"]["boxes"], shares
