"""GPU parity: projection build (create_projection_mat), camera weights and skipped
nodes vs the CPU oracle.  Bar: identical pixel index per node, bit-identical uv."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def setup_case(oracle, mesh, cam_kw, size):
    from upsp_processing_amd import _capi, synthetic as syn
    v, t = mesh
    s9, tn = syn.soup(v, t)
    nrm = syn.node_normals(v, t)
    c = syn.pinhole_camera(size[0], size[1], **cam_kw)
    cam_g = _capi.make_camera(c["K"], c["dist"], c["R"], c["t"], *size)
    cam_o = oracle.make_camera(c["K"], c["dist"], c["R"], c["t"], *size)
    return v, t, s9, tn, nrm, cam_g, cam_o


@pytest.mark.parametrize("case", ["sphere10k", "tunnel", "tunnel_k1"])
def test_projection_matches_oracle(gpu_lib, oracle, case):
    from upsp_processing_amd import engine, synthetic as syn
    if case == "sphere10k":      # BASELINE config 1: 512x512 frame, 10k-tri sphere
        mesh, kw, size = syn.uv_sphere(50, 100), dict(), (512, 512)
    elif case == "tunnel":
        mesh, kw, size = syn.tunnel_model(100, 240, 40, 80), dict(center=(0, 0, 20), half_extent=6.0), (1024, 512)
    else:
        mesh, kw, size = syn.tunnel_model(60, 120, 24, 48), dict(center=(0, 0, 20), half_extent=5.0, k1=-0.09, azimuth_deg=35), (640, 480)
    v, t, s9, tn, nrm, cam_g, cam_o = setup_case(oracle, mesh, kw, size)
    bvh = engine.BVH(s9)
    obv = oracle.OracleBVH(s9)
    dn = (np.arange(v.shape[0]) % 11 != 0).astype(np.uint8) if case == "tunnel" else None
    g = engine.build_projection(bvh, cam_g, v, nrm, tn, 70.0, datanode=dn, nodecount=True)
    o = oracle.create_projection(obv, cam_o, v, nrm, tn, engine.oblique_threshold(70.0), datanode=dn)
    pix = g["pix"].cpu().numpy()
    assert (pix >= 0).sum() > 100
    assert np.array_equal(pix, o["pix"])
    assert np.array_equal(g["uv"].cpu().numpy().view(np.int32), o["uv"].view(np.int32))
    assert g["nrays"] == o["nrays"]
    assert np.array_equal(g["nodecount"].cpu().numpy(), o["nodecount"])
    sk, cnt = engine.skipped_nodes(g["pix"])
    assert np.array_equal(sk.cpu().numpy(), oracle.skipped_nodes(o["pix"])) and cnt == (pix < 0).sum()
    # same build with the node -> triangle adjacency handed to the BVH (bounded visibility
    # rays, upsp_bvh_set_tri_nodes): identical verdicts, identical reference ray count
    import torch
    d_tn = torch.as_tensor(tn).cuda()
    bvh.set_tri_nodes(d_tn, v.shape[0])
    bvh.enable_stats(True)
    g2 = engine.build_projection(bvh, cam_g, v, nrm, d_tn, 70.0, datanode=dn, nodecount=True)
    st_bounded = bvh.last_stats()
    assert np.array_equal(g2["pix"].cpu().numpy(), o["pix"])
    assert np.array_equal(g2["uv"].cpu().numpy().view(np.int32), o["uv"].view(np.int32))
    assert g2["nrays"] == o["nrays"]
    g3 = engine.build_projection(bvh, cam_g, v, nrm, tn, 70.0, datanode=dn)   # other buffer: classic path
    st_classic = bvh.last_stats()
    assert np.array_equal(g3["pix"].cpu().numpy(), o["pix"])
    assert st_bounded["nodes"] <= st_classic["nodes"]                         # the bound only ever prunes
    # counts=False (what the frame loops use): the oblique test runs before the rays and the nodes it rejects cast
    # none -- same entries, uv and node-count image; fewer rays than the reference casts
    bvh.enable_stats(False)
    for tnn in (d_tn, tn):
        g4 = engine.build_projection(bvh, cam_g, v, nrm, tnn, 70.0, datanode=dn, nodecount=True, counts=False)
        assert np.array_equal(g4["pix"].cpu().numpy(), o["pix"])
        assert np.array_equal(g4["uv"].cpu().numpy().view(np.int32), o["uv"].view(np.int32))
        assert np.array_equal(g4["nodecount"].cpu().numpy(), o["nodecount"])
        pc = engine.projection_counts(bvh)
        import os
        if os.environ.get("UPSP_OBLIQUE_CULL", "1") == "1":
            assert (pix >= 0).sum() <= pc["primary_rays"] < g["primary_rays"] and pc["nrays"] < o["nrays"]


@pytest.mark.parametrize("steps,stack", [(8, 4096), (8, 128), (40, 4096), (1, 130)])
def test_heavy_ray_handoff(gpu_lib, oracle, monkeypatch, steps, stack):
    """Rays that need more than UPSP_HEAVY_STEPS node visits + triangle tests leave the one-lane traversal and are
    walked by a whole wave (heavy_kernel; default 256 steps: only the rays through high-valence vertices).  With a
    threshold of a few steps most rays of the build take that road, primary and retries; with a 128-entry stack
    most of those overflow it and take the one-lane fallback inside heavy_kernel.  Same verdicts either way."""
    import torch
    from upsp_processing_amd import engine, synthetic as syn
    mesh, kw, size = syn.tunnel_model(100, 240, 40, 80), dict(center=(0, 0, 20), half_extent=6.0), (1024, 512)
    v, t, s9, tn, nrm, cam_g, cam_o = setup_case(oracle, mesh, kw, size)
    bvh = engine.BVH(s9)
    obv = oracle.OracleBVH(s9)
    o = oracle.create_projection(obv, cam_o, v, nrm, tn, engine.oblique_threshold(70.0))
    monkeypatch.setenv("UPSP_HEAVY_STEPS", str(steps))
    monkeypatch.setenv("UPSP_HEAVY_STACK", str(stack))
    d_tn = torch.as_tensor(tn).cuda()
    for adjacency in (False, True):
        if adjacency:
            bvh.set_tri_nodes(d_tn, v.shape[0])
        g = engine.build_projection(bvh, cam_g, v, nrm, d_tn if adjacency else tn, 70.0)
        assert np.array_equal(g["pix"].cpu().numpy(), o["pix"])
        assert np.array_equal(g["uv"].cpu().numpy().view(np.int32), o["uv"].view(np.int32))
        assert g["nrays"] == o["nrays"]
        g = engine.build_projection(bvh, cam_g, v, nrm, d_tn if adjacency else tn, 70.0, counts=False)
        assert np.array_equal(g["pix"].cpu().numpy(), o["pix"])
    # a pole of the UV sphere seen head-on: the ray to the pole vertex meets every triangle of its fan
    mesh = syn.uv_sphere(60, 900)
    from upsp_processing_amd import _capi
    v, t = mesh
    s9, tn = syn.soup(v, t)
    nrm = syn.node_normals(v, t)
    bvh, obv = engine.BVH(s9), oracle.OracleBVH(s9)
    for az in (0.0, 90.0, 37.0):
        c = syn.pinhole_camera(256, 256, center=(0, 0, 20), half_extent=1.5, azimuth_deg=az)
        cg = _capi.make_camera(c["K"], c["dist"], c["R"], c["t"], 256, 256)
        co = oracle.make_camera(c["K"], c["dist"], c["R"], c["t"], 256, 256)
        g = engine.build_projection(bvh, cg, v, nrm, tn, 70.0)
        o = oracle.create_projection(obv, co, v, nrm, tn, engine.oblique_threshold(70.0))
        assert np.array_equal(g["pix"].cpu().numpy(), o["pix"])
        assert g["nrays"] == o["nrays"]


def test_dense_soup_every_ray_heavy(gpu_lib, oracle):
    """A soup of large overlapping triangles (the kind-0 scenes of tests/debug/soak_raycast.py with spread 2): every
    camera -> node ray needs ~1000 node visits and ~800 triangle tests, so every primary ray and every retry of the build
    is handed to heavy_kernel (tens of thousands of work items), and the batch queries to heavy_cast_kernel."""
    import os
    import torch
    from upsp_processing_amd import _capi, engine, synthetic as syn
    rng = np.random.default_rng(21150)
    n = 1400
    c = rng.normal(size=(n, 1, 3)) * 3
    s9 = (c + rng.normal(size=(n, 3, 3)) * 2.0).astype(np.float32).reshape(-1)
    v = np.ascontiguousarray(s9.reshape(-1, 3), np.float32)
    tn = np.arange(v.shape[0], dtype=np.int32)
    nrm = np.tile(np.float32([0, 0, 1]), (v.shape[0], 1))
    nrm[::2] = np.float32([0, 0, -1])            # half of the nodes pass the oblique test for a camera on +z
    bvh, obv = engine.BVH(s9), oracle.OracleBVH(s9)
    cd = syn.pinhole_camera(256, 256, center=(1.0, -2.0, 40.0), half_extent=12.0)
    cg = _capi.make_camera(cd["K"], cd["dist"], cd["R"], cd["t"], 256, 256)
    co = oracle.make_camera(cd["K"], cd["dist"], cd["R"], cd["t"], 256, 256)
    want = oracle.create_projection(obv, co, v, nrm, tn, engine.oblique_threshold(70.0), threads=os.cpu_count() or 1)
    assert want["nrays"] > 3 * v.shape[0]        # most nodes go through the retries
    d_tn = torch.as_tensor(tn).cuda()
    for adjacency in (False, True):
        if adjacency:
            bvh.set_tri_nodes(d_tn, v.shape[0])
        g = engine.build_projection(bvh, cg, v, nrm, d_tn, 70.0)
        assert np.array_equal(g["pix"].cpu().numpy(), want["pix"]) and g["nrays"] == want["nrays"]
        g = engine.build_projection(bvh, cg, v, nrm, d_tn, 70.0, counts=False)
        assert np.array_equal(g["pix"].cpu().numpy(), want["pix"])
    cam = oracle.cam_center(co).astype(np.float32)
    d = (v - cam).astype(np.float32)
    gh, oh = bvh.intersect(cam, d, want=("hit", "t", "prim")), obv.intersect(cam, d, threads=os.cpu_count() or 1)
    assert np.array_equal(gh["prim"].cpu().numpy(), oh["prim"])
    assert np.array_equal(gh["t"].cpu().numpy().view(np.int32), oh["t"].view(np.int32))
    assert np.array_equal(bvh.occluded(cam, d).cpu().numpy(), oh["hit"])


def test_multi_camera_weights(gpu_lib, oracle):
    from upsp_processing_amd import engine, _capi, synthetic as syn
    v, t = syn.tunnel_model(60, 120, 24, 48)
    s9, tn = syn.soup(v, t)
    nrm = syn.node_normals(v, t)
    bvh, obv = engine.BVH(s9), oracle.OracleBVH(s9)
    pix_g, pix_o, centers = [], [], []
    for az in (0, 60, 120, 200):
        c = syn.pinhole_camera(512, 512, center=(0.37 * az / 60, 0.2, 20 + az / 50), half_extent=6.0,
                               azimuth_deg=az)
        cg = _capi.make_camera(c["K"], c["dist"], c["R"], c["t"], 512, 512)
        co = oracle.make_camera(c["K"], c["dist"], c["R"], c["t"], 512, 512)
        pix_g.append(engine.build_projection(bvh, cg, v, nrm, tn, 70.0)["pix"])
        pix_o.append(oracle.create_projection(obv, co, v, nrm, tn, engine.oblique_threshold(70.0))["pix"])
        centers.append(engine.camera_center(cg))
    import torch
    pg = torch.stack(pix_g)
    po = np.stack(pix_o)
    assert np.array_equal(pg.cpu().numpy(), po)
    assert ((po >= 0).sum(0) >= 2).sum() > 50      # overlap exists
    # angle between (node - camera) and the normal, as angle_between() computes it
    ang = np.stack([np.arccos(np.clip(((v - np.asarray(c, np.float32)) * nrm).sum(1).astype(np.float64)
                    / np.linalg.norm((v - np.asarray(c, np.float32)).astype(np.float64), axis=1)
                    / np.linalg.norm(nrm.astype(np.float64), axis=1), -1, 1)) for c in centers])
    ang = np.where(po >= 0, ang, -1.0)
    top2 = np.sort(ang, axis=0)[-2:]
    ambiguous = (top2[1] - top2[0]) < 1e-6          # BestView argmax decided by the last ulp of acos
    for mode, m in (("best_view", 0), ("average_view", 1)):
        wg = engine.projection_weights(pg, v, nrm, np.array(centers), mode).cpu().numpy()
        wo = oracle.adjust_weights(po, np.ones_like(po, dtype=np.float32), v, nrm, np.array(centers), m)
        ok = ~ambiguous if m == 0 else np.ones_like(ambiguous)
        assert ambiguous.mean() < 0.5, ambiguous.mean()
        # f64 acos -> f32: 1 ulp tolerance
        assert np.allclose(wg[:, ok], wo[:, ok], rtol=2e-7, atol=0)
        seen = po >= 0
        s = (wg * seen).sum(0)
        assert np.allclose(s[seen.any(0)], 1.0, atol=1e-6)


def test_empty_and_ragged(gpu_lib):
    from upsp_processing_amd import engine, _capi, synthetic as syn
    v, t = syn.uv_sphere(8, 16)
    s9, tn = syn.soup(v, t)
    bvh = engine.BVH(s9)
    c = syn.pinhole_camera(64, 48)
    cam = _capi.make_camera(c["K"], c["dist"], c["R"], c["t"], 64, 48)
    # principal point far outside the sensor: nothing projects into the frame
    c2 = syn.pinhole_camera(64, 48, center=(0, 0, 4))
    c2["K"][0, 2] += 10000.0
    cam2 = _capi.make_camera(c2["K"], c2["dist"], c2["R"], c2["t"], 64, 48)
    g = engine.build_projection(bvh, cam2, v, syn.node_normals(v, t), tn, 70.0)
    assert (g["pix"].cpu().numpy() == -1).all()
    with pytest.raises(ValueError):
        engine.build_projection(bvh, cam, v, v, tn[:-3], 70.0)


def test_far_off_distorted_nodes(gpu_lib, oracle):
    """Nodes whose distorted projection is beyond the int range are out of frame on both sides
    (cvRound saturation), with and without the adjacency / occluder-witness path."""
    import torch
    import refdata
    from upsp_processing_amd import _capi, engine, synthetic as syn
    v, t, c, (W, H) = refdata.distorted_plates_scene()
    s9, tn = syn.soup(v, t)
    nrm = syn.node_normals(v, t)
    cam_g = _capi.make_camera(c["K"], c["dist"], c["R"], c["t"], W, H)
    cam_o = oracle.make_camera(c["K"], c["dist"], c["R"], c["t"], W, H)
    o = oracle.create_projection(oracle.OracleBVH(s9), cam_o, v, nrm, tn, engine.oblique_threshold(70.0))
    bvh = engine.BVH(s9)
    d_tn = torch.as_tensor(tn).cuda()
    for adj in (False, True):
        if adj:
            bvh.set_tri_nodes(d_tn, v.shape[0])
        g = engine.build_projection(bvh, cam_g, v, nrm, d_tn, 70.0)
        assert np.array_equal(g["pix"].cpu().numpy(), o["pix"])
        assert g["nrays"] == o["nrays"] == 35


@pytest.mark.parametrize("model", ["quad", "uv"])
def test_projection_full_size_vs_oracle(gpu_lib, oracle, model):
    """BASELINE configs[1] at its own size: the projection of the bench's 1 001 904-triangle tunnel
    model (and of the UV-sphere variant with 1000-valent polar fans) onto the 1024 x 1024 frame,
    `pix`, `uv`, node count image and the reference ray count against the oracle (OpenMP; 0.3 s on
    the GPU box's host cores), with and without the node -> triangle adjacency."""
    import os
    import torch
    from upsp_processing_amd import engine, synthetic as syn
    mesh = syn.tunnel_model_quad() if model == "quad" else syn.tunnel_model()
    v, t, s9, tn, nrm, cam_g, cam_o = setup_case(oracle, mesh, dict(center=(0, 0, 20), half_extent=6.0), (1024, 1024))
    assert t.shape[0] > 1_000_000
    bvh = engine.BVH(s9)
    obv = oracle.OracleBVH(s9)
    o = oracle.create_projection(obv, cam_o, v, nrm, tn, engine.oblique_threshold(70.0), threads=os.cpu_count() or 1)
    d_tn = torch.as_tensor(tn).cuda()
    g0 = engine.build_projection(bvh, cam_g, v, nrm, tn, 70.0, nodecount=True)      # classic retries
    bvh.set_tri_nodes(d_tn, v.shape[0])
    g1 = engine.build_projection(bvh, cam_g, v, nrm, d_tn, 70.0, nodecount=True)    # own-triangle bound + witness
    assert (o["pix"] >= 0).sum() > 100_000
    for g in (g0, g1):
        assert np.array_equal(g["pix"].cpu().numpy(), o["pix"])
        assert np.array_equal(g["uv"].cpu().numpy().view(np.int32), o["uv"].view(np.int32))
        assert g["nrays"] == o["nrays"]
        assert np.array_equal(g["nodecount"].cpu().numpy(), o["nodecount"])
    bvh.close()


def test_shared_tree_builds_side_by_side(gpu_lib):
    """engine.BVH.share (upsp_bvh_share): handles on one tree with query scratch of their own.  The projection builds of four
    cameras queued at the same time on four streams, one handle each, give the entries and uv of the builds run one after the
    other on the owner; a batch query on a share gives the owner's hits; the adjacency cannot be set on a share; the shares
    outlive nothing (closing them leaves the owner usable)."""
    import torch
    from upsp_processing_amd import _capi, engine, synthetic as syn
    v, t = syn.tunnel_model(100, 240, 40, 80)
    s9, tn = syn.soup(v, t)
    nrm = syn.node_normals(v, t)
    size = 512
    cams = []
    for c in range(4):
        cd = syn.pinhole_camera(size, size, center=(0, 0, 20), half_extent=6.0, azimuth_deg=90.0 * c)
        cams.append(_capi.make_camera(cd["K"], cd["dist"], cd["R"], cd["t"], size, size))
    bvh = engine.BVH(s9)
    d_v, d_n, d_tn = [torch.as_tensor(x).cuda() for x in (v, nrm, tn)]
    bvh.set_tri_nodes(d_tn, v.shape[0])
    want = [engine.build_projection(bvh, cam, d_v, d_n, d_tn, 70.0, counts=False) for cam in cams]
    torch.cuda.synchronize()
    handles = [bvh] + [bvh.share() for _ in cams[1:]]
    streams = [torch.cuda.Stream() for _ in cams]
    for rep in range(3):                       # (repeated: the scratch of every handle is reused)
        got = []
        for h, cam, st in zip(handles, cams, streams):
            with torch.cuda.stream(st):
                got.append(engine.build_projection(h, cam, d_v, d_n, d_tn, 70.0, counts=False))
        torch.cuda.synchronize()
        for g, w in zip(got, want):
            assert (w["pix"] >= 0).sum() > 1000
            assert torch.equal(g["pix"], w["pix"]) and torch.equal(g["uv"].view(torch.int32), w["uv"].view(torch.int32))
    for h in handles:
        h.check()
    rng = np.random.default_rng(2)
    org = rng.normal(size=(5000, 3)).astype(np.float32) * 8
    dirs = -org + rng.normal(size=(5000, 3)).astype(np.float32)
    a = bvh.intersect(org, dirs, want=("hit", "t", "prim"))
    b = handles[2].intersect(org, dirs, want=("hit", "t", "prim"))
    hit = a["hit"] != 0
    assert int(hit.sum()) > 100 and torch.equal(a["hit"], b["hit"])
    assert torch.equal(a["prim"][hit], b["prim"][hit]) and torch.equal(a["t"][hit].view(torch.int32), b["t"][hit].view(torch.int32))
    with pytest.raises(Exception):
        handles[1].set_tri_nodes(d_tn, v.shape[0])
    for h in handles[1:]:
        h.close()
    again = engine.build_projection(bvh, cams[0], d_v, d_n, d_tn, 70.0, counts=False)
    assert torch.equal(again["pix"], want[0]["pix"])


def test_length_binned_primary_lists_same_projection(gpu_lib, oracle, monkeypatch):
    """UPSP_RAY_BINS=1 (round 6, off by default: measured slower): a REPEATED build bins the dense list of primary rays by the step
    counts of the build before it (six lists, an eighth of every bin per XCD).  Ordering only: pix / uv of the first build (plain
    list), the second and third (binned) and of a build with the switch off are identical, and the oracle's."""
    import torch
    from upsp_processing_amd import _capi, engine, synthetic as syn
    v, t = syn.tunnel_model(60, 120, 24, 48)
    s9, tn = syn.soup(v, t)
    nrm = syn.node_normals(v, t)
    c = syn.pinhole_camera(512, 512, center=(0, 0, 20), half_extent=6.0, fill=0.7)
    cam = _capi.make_camera(c["K"], c["dist"], c["R"], c["t"], 512, 512)
    cam_o = oracle.make_camera(c["K"], c["dist"], c["R"], c["t"], 512, 512)
    o = oracle.create_projection(oracle.OracleBVH(s9), cam_o, v, nrm, tn, engine.oblique_threshold(70.0))
    bvh = engine.BVH(s9)
    d_v, d_n, d_tn = torch.as_tensor(v).cuda(), torch.as_tensor(nrm).cuda(), torch.as_tensor(tn).cuda()
    bvh.set_tri_nodes(d_tn, v.shape[0])
    monkeypatch.setenv("UPSP_RAY_BINS", "1")
    outs = [engine.build_projection(bvh, cam, d_v, d_n, d_tn, 70.0, counts=False) for _ in range(3)]
    counted = engine.build_projection(bvh, cam, d_v, d_n, d_tn, 70.0, counts=True)      # the reference's order, binned too
    monkeypatch.setenv("UPSP_RAY_BINS", "0")
    outs.append(engine.build_projection(bvh, cam, d_v, d_n, d_tn, 70.0, counts=False))
    for g in outs + [counted]:
        assert np.array_equal(g["pix"].cpu().numpy(), o["pix"])
        assert np.array_equal(g["uv"].cpu().numpy().view(np.int32).reshape(-1), o["uv"].view(np.int32).reshape(-1))
    assert counted["nrays"] == o["nrays"]
    bvh.check()
