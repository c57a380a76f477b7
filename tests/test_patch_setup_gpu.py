"""Phase-0 patch set-up on the GPU engine (visibility ray cast + nearest node + projection in
the library, pixel lists on the host) against the C oracle, on a synthetic model and on the
reference's own data set (fml_tc3_volume.grid + camera01 + fml_tc3_volume.tgts).
Index / pixel-list results: bit-exact; diameters: float, bit-exact (same arithmetic)."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _write_targets(path, xyz, diam, fid_from):
    with open(path, "w") as f:
        f.write("#GRID_FILE: synthetic\n*Targets\n")
        for i in range(fid_from):
            f.write("%4d %10.4f %10.4f %10.4f 0 0 1 %6.3f 1 1 1 st%02d\n" % (i + 1, *xyz[i], diam[i], i + 1))
        f.write("*Taps\n   1 0.0 0.0 0.0 0 0 1 0.02 9 1 1 CP01\n*Fiducials\n")
        for i in range(fid_from, len(xyz)):
            f.write("%4d %10.4f %10.4f %10.4f 0 0 1 %6.3f 1 1 1 fd%02d\n" % (i + 1, *xyz[i], diam[i], i + 1))


def _compare(oracle, ps, bvh_g, bvh_o, kd, cam_g, cam_o, size, nodes, normals, d_nodes, tfile, frame,
             oblique=70.0):
    targs = ps.read_psp_target_file(tfile) + ps.read_psp_target_file(tfile, "*Fiducials")
    xyz = np.stack([t.xyz for t in targs])
    thr = ps.target_oblique_threshold(oblique)
    vis = ps.get_targets(bvh_g, cam_g, size, targs, d_nodes, normals, thr)
    keep = oracle.get_targets(bvh_o, kd, cam_o, normals, xyz, thr)
    assert [t.num for t in vis] == [targs[i].num for i in np.nonzero(keep)[0]]
    ps.map_points_to_image(cam_g, vis)
    uv = np.stack([t.uv for t in vis]) if vis else np.zeros((0, 2), np.float32)
    uv_o = (oracle.project_points(cam_o, np.stack([t.xyz for t in vis])) if vis else np.zeros((0, 2), np.float32))
    assert np.array_equal(uv, uv_o)
    d_in = np.array([t.diameter for t in vis], np.float32)
    diams = ps.get_target_diameters(cam_g, size, vis, d_nodes, normals)
    want = oracle.target_diameters(kd, cam_o, normals, np.stack([t.xyz for t in vis]), uv, d_in)
    assert np.array_equal(diams, want) and (diams > 0).any()
    patches, vis2, thresh = ps.initialize_image_patches(bvh_g, cam_g, size, tfile, frame, d_nodes, normals,
                                                        oblique_angle=oblique)
    sf = np.float32(1.2)
    d_sc = (want * sf).astype(np.float32)
    order, off = oracle.cluster_points(uv, d_sc, 3)
    e, c = oracle.intensity_histc(frame, 12, 256)
    t_o = int(e[oracle.first_min_threshold(c, 5)]) + 5
    assert thresh == t_o
    tab = oracle.patch_tables(uv, d_sc, order, off, size, 2, 1, ref=frame, thresh=t_o, offset=2)
    assert len(tab) == len(patches)
    for g, w in zip(patches, tab):
        for k in ("ix", "iy", "bx", "by"):
            assert np.array_equal(g[k], w[k]), k
    return vis, patches


def test_patch_setup_synthetic(gpu_lib, oracle, tmp_path):
    import torch
    from upsp_processing_amd import _capi, engine, patch_setup as ps, synthetic as syn
    W, H = 512, 384
    v, t = syn.tunnel_model_quad(40, 14)
    s9, tn = syn.soup(v, t)
    nrm = syn.node_normals(v, t)
    c = syn.pinhole_camera(W, H, center=(0.2, 0.1, 20), half_extent=6.5)
    cam_g = _capi.make_camera(c["K"], c["dist"], c["R"], c["t"], W, H)
    cam_o = oracle.make_camera(c["K"], c["dist"], c["R"], c["t"], W, H)
    rng = np.random.default_rng(12)
    pick = rng.choice(len(v), 60, replace=False)          # targets ON the surface, all around the model
    xyz = v[pick] * np.float32(1.0005)
    pairs = pick[:6]                                      # close neighbours -> multi-target clusters
    xyz = np.concatenate([xyz, v[pairs] * np.float32(1.0005) + np.array([0.12, 0.05, 0], np.float32)])
    diam = np.full(len(xyz), 0.08, np.float32)
    diam[5] = 0.0
    tfile = str(tmp_path / "syn.tgts")
    _write_targets(tfile, xyz, diam, fid_from=50)
    frame = syn.synth_frames_numpy(1, H, W, seed=5, noise=3.0)[0]
    frame[:40] = 60                                       # dark background band -> bimodal histogram
    bvh_g, bvh_o, kd = engine.BVH(s9), oracle.OracleBVH(s9), oracle.KdTree(v)
    d_nodes = torch.as_tensor(v).cuda()
    vis, patches = _compare(oracle, ps, bvh_g, bvh_o, kd, cam_g, cam_o, (W, H), v, nrm, d_nodes, tfile, frame)
    assert 5 < len(vis) < len(xyz)                        # some are hidden / oblique / behind
    assert any(p["ix"].size for p in patches)
    bvh_g.close()


def test_patch_setup_reference_dataset(gpu_lib, oracle, fml):
    """The reference's own model, calibration and target list."""
    import torch
    from upsp_processing_amd import _capi, engine, patch_setup as ps
    W, H = 1024, 512
    cam_g = _capi.make_camera(fml["cm"], fml["dist"], fml["rmat"], fml["tvec"], W, H)
    cam_o = oracle.make_camera(fml["cm"], fml["dist"], fml["rmat"], fml["tvec"], W, H)
    nodes = fml["nodes"].astype(np.float32)
    # the fixture holds un-normalised first-face normals (Python visibility checker); the model
    # normals psp_process uses are unit vectors (calcNormals)
    n = fml["norms"].astype(np.float64)
    mag = np.linalg.norm(n, axis=1, keepdims=True)
    normals = np.where(mag == 0, n, n / np.where(mag == 0, 1, mag)).astype(np.float32)
    bvh_g, bvh_o, kd = engine.BVH(fml["prims"]), oracle.OracleBVH(fml["prims"]), oracle.KdTree(nodes)
    rng = np.random.default_rng(8)
    frame = (1600 + 150 * rng.standard_normal((H, W))).clip(0, 4095).astype(np.uint16)
    frame[:, :200] = (80 + 10 * rng.standard_normal((H, 200))).clip(0, 4095).astype(np.uint16)
    d_nodes = torch.as_tensor(nodes).cuda()
    vis, patches = _compare(oracle, ps, bvh_g, bvh_o, kd, cam_g, cam_o, (W, H), nodes, normals, d_nodes,
                            os.path.join(GOLD, "fml_tc3_volume.tgts"), frame)
    assert len(vis) >= 10
    bvh_g.close()
