"""CPU: pin the oracle (oracle/*.c) on the reference's own known-answer tests for the
ray-cast boundary -- test/python/test_visibility.py (toy scenes :256-322, camera01
regression :243-254, face / node counts :227-241) -- before it is trusted as the
checker of the HIP path."""
import numpy as np
import pytest

from upsp_processing_amd.visibility import VisibilityChecker


class OracleScene:
    """BVH-shaped adapter over the oracle so the VisibilityChecker mirror can drive it."""

    def __init__(self, orc, prims):
        self.bvh = orc.OracleBVH(prims)

    def occluded_many(self, o, d):
        return self.bvh.intersect(o, d)["hit"]

    def intersect_many(self, o, d):
        return self.bvh.intersect(o, d)


@pytest.fixture(scope="module")
def toy(oracle):
    t0 = [0., 0., 1., 0., 1., 0., 1., 0., 0.]
    t1 = [0., 0., 1., 0., 1., 0., 0., 1., 1.]
    return VisibilityChecker(OracleScene(oracle, np.array(t0 + t1, dtype=np.float32)),
                             oblique_angle=70, epsilon=1e-4)


def test_update_oblique_angle(toy):
    assert toy.squared_cos_angle == np.cos(np.deg2rad(70)) ** 2
    toy.update_oblique_angle(40)
    assert toy.squared_cos_angle == np.cos(np.deg2rad(40)) ** 2
    toy.update_oblique_angle(70)


def test_back_facing(toy):
    cam = np.array([1., 1., 1.]).reshape(3, 1)
    nodes = np.array([[2., 2., 2.], [3., 3., 3.]])
    normals = np.array([[1., 1., 1.], [1., 1., 1.]])
    assert toy.is_visible(cam, nodes, normals).tolist() == []


def test_occluded(toy):
    cam = np.array([1., 1., 1.]).reshape(3, 1)
    nodes = np.array([[0., 0., 0.], [-1., -1., -1.]])
    normals = np.array([[0.9, 0.9, 0.9], [1.1, 1.1, 1.1]])
    assert toy.is_visible(cam, nodes, normals).tolist() == []


def test_all_visible(toy):
    cam = np.array([-8., -8., 0.])
    nodes = np.array([[-5., -5., 0.], [-5., -1., 0]])
    normals = np.array([[-0.9, 0., 0.], [0., -1.2, 0.]])
    assert toy.is_visible(cam, nodes, normals).tolist() == [0, 1]
    cam = np.array([-2., -2., -2.]).reshape(3, 1)
    nodes = np.array([[-1., -1., -1.], [0., 0., 0.]])
    normals = np.array([[-1., -1., -1.], [-1., -1., -1.]])
    assert toy.is_visible(cam, nodes, normals).tolist() == [0, 1]


def test_some_visible(toy):
    cam = np.array([1., 1., 1.]).reshape(3, 1)
    nodes = np.array([[2., 2., 2.], [3., 3., 3.], [0.9, 0.9, 0.9], [0.5, 0.5, 0.5],
                      [0., 0., 0.], [-1., -1., -1.]])
    normals = np.ones((6, 3))
    assert toy.is_visible(cam, nodes, normals).tolist() == [2, 3]


def test_grid_counts(fml):
    # test_get_faces_and_face_normals / test_get_tvecs_and_norms
    assert fml["nfaces"] == 609120
    assert fml["nodes"].shape == (304566, 3)
    assert fml["norms"].shape == (304566, 3)
    assert fml["prims"].size == 609120 * 9


def test_camera01_regression(oracle, fml):
    """test_camera01: 148 608 of 304 566 nodes visible from camera01."""
    vc = VisibilityChecker(OracleScene(oracle, fml["prims"]), oblique_angle=70, epsilon=1e-4)
    vis = vc.is_visible(fml["cam_t"], fml["nodes"], fml["norms"])
    assert len(vis) == 148608
    golden = np.load(__import__("os").path.join(__import__("refdata").GOLDEN, "camera01_visible.npz"))
    assert np.array_equal(vis, golden["visible"])


def test_bvh_structure(oracle):
    """LinearNode invariants of the flattened tree (pspRT.cpp:433-454)."""
    from upsp_processing_amd import synthetic as syn
    v, t = syn.uv_sphere(20, 40)
    s9, _ = syn.soup(v, t)
    bvh = oracle.OracleBVH(s9)
    nodes = bvh.nodes()
    leaves = nodes[nodes["nprims"] > 0]
    assert leaves["nprims"].sum() == t.shape[0]
    assert leaves["nprims"].max() <= 4
    assert sorted(bvh.prim_ids().tolist()) == list(range(t.shape[0]))
    # interior node i has its first child at i+1 and bounds = union of children
    for i in np.nonzero(nodes["nprims"] == 0)[0][:200]:
        a, b = nodes[i + 1], nodes[nodes[i]["offset"]]
        assert np.array_equal(nodes[i]["bmin"], np.minimum(a["bmin"], b["bmin"]))
        assert np.array_equal(nodes[i]["bmax"], np.maximum(a["bmax"], b["bmax"]))


def test_closest_hit_is_brute_force_minimum(oracle):
    """BVH traversal == brute force over all triangles with the same triangle test."""
    import ctypes as C
    from upsp_processing_amd import synthetic as syn
    v, t = syn.uv_sphere(12, 24)
    s9, _ = syn.soup(v, t)
    bvh = oracle.OracleBVH(s9)
    rng = np.random.default_rng(5)
    org = rng.normal(size=(300, 3)).astype(np.float32) * 3
    dirs = (-org + rng.normal(size=(300, 3)) * 0.7).astype(np.float32)
    res = bvh.intersect(org, dirs)
    L = oracle.lib()
    tri = s9.reshape(-1, 9)
    for i in range(300):
        r = oracle.Ray()
        L.orc_ray_init(C.byref(r), org[i].ctypes.data_as(C.c_void_p), dirs[i].ctypes.data_as(C.c_void_p))
        best, bp = np.float32(np.finfo(np.float32).max), -1
        for k in range(tri.shape[0]):
            h = oracle.Hit()
            L.orc_hit_init(C.byref(h))
            L.orc_tri_intersect.restype = C.c_int
            if L.orc_tri_intersect(C.byref(r), tri[k, 0:3].ctypes.data_as(C.c_void_p),
                                   tri[k, 3:6].ctypes.data_as(C.c_void_p),
                                   tri[k, 6:9].ctypes.data_as(C.c_void_p), k, C.byref(h)):
                if h.t < best:
                    best, bp = np.float32(h.t), k
        assert res["hit"][i] == (bp >= 0)
        if bp >= 0:
            assert res["t"][i] == best


def test_hot_pixels_oracle(oracle):
    img = np.full((8, 10), 1000, np.uint16)
    img[3, 4] = 4095
    img[0, 0] = 4090          # corner: 2 neighbours
    img[7, 9] = 4064
    out, st = oracle.fix_hot_pixels(img)
    assert st == 3 and out[3, 4] == 1000 and out[0, 0] == 1000 and out[7, 9] == 1000
    img2 = img.copy()
    img2[5, 5:8] = 4095       # 6 hot pixels > max_hot=5 -> untouched
    out2, st2 = oracle.fix_hot_pixels(img2)
    assert st2 == -1 and np.array_equal(out2, img2)
    img3 = np.full((4, 4), 4000, np.uint16)
    img3[1, 1] = 4095         # change 95 <= 512 -> kept
    out3, st3 = oracle.fix_hot_pixels(img3)
    assert st3 == 0 and out3[1, 1] == 4095


def test_apportion_and_transpose(oracle):
    st, ex = oracle.apportion(10, 4)
    assert st.tolist() == [0, 3, 6, 8] and ex.tolist() == [3, 3, 2, 2]
    st, ex = oracle.apportion(3, 5)
    assert ex.tolist() == [1, 1, 1, 0, 0]
    a = np.arange(12, dtype=np.float32).reshape(3, 4)
    assert np.array_equal(oracle.transpose(a), a.T)


def test_projection_oracle_sphere(oracle):
    """create_projection_mat restatement on the config-1 plumbing case (512x512, 10k-tri sphere)."""
    from upsp_processing_amd import synthetic as syn
    v, t = syn.uv_sphere(50, 100)
    s9, tn = syn.soup(v, t)
    nrm = syn.node_normals(v, t)
    bvh = oracle.OracleBVH(s9)
    c = syn.pinhole_camera(512, 512)
    cam = oracle.make_camera(c["K"], c["dist"], c["R"], c["t"], 512, 512)
    thr = np.float32((180.0 - 70.0) * 3.141592653589793 / 180.0)
    r = oracle.create_projection(bvh, cam, v, nrm, tn, thr)
    vis = r["pix"] >= 0
    # the camera sits at z=+4: accepted nodes are on the +z cap within the oblique cone
    assert vis.sum() == r["accepted"] > 1000
    assert (v[vis, 2] > 0.3).all()
    # every accepted node lands on the pixel nearest to its projection
    uv = oracle.project_points(cam, v[vis])
    assert np.array_equal(r["pix"][vis], np.round(uv[:, 1]).astype(int) * 512 + np.round(uv[:, 0]).astype(int))
    assert r["nodecount"].sum() == vis.sum()
    sk = oracle.skipped_nodes(r["pix"])
    assert np.array_equal(sk, ~vis)


def test_fixture_manifest(fml):
    """Inputs prepared by tests/refdata.py hash to what tests/golden/make_golden.py recorded
    when it ran the reference's own Python input preparation on the same files."""
    import hashlib, json, os
    import refdata
    man = json.load(open(os.path.join(refdata.GOLDEN, "golden_manifest.json")))
    sha = lambda a: hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()
    assert sha(fml["prims"]) == man["primitives_sha256"]
    assert sha(fml["nodes"]) == man["nodes_sha256"]
    assert sha(fml["norms"]) == man["normals_sha256"]
    vis = np.load(os.path.join(refdata.GOLDEN, "camera01_visible.npz"))["visible"]
    assert len(vis) == man["visible_count"] == 148608 and sha(vis) == man["visible_sha256"]


def test_far_off_distorted_nodes_are_out_of_frame(oracle):
    """cv::Point2f -> Point2i is cvRound = cvtss2si: a node whose distorted projection lies beyond
    the int range (|pt| ~ 1e10 px) gives 0x80000000 -- out of frame -- and must not wrap around
    into the frame (psp_process.cpp:252, upsp::contains(Size, Point2i(pt)))."""
    import refdata
    from upsp_processing_amd import synthetic as syn
    v, t, c, (W, H) = refdata.distorted_plates_scene()
    s9, tn = syn.soup(v, t)
    nrm = syn.node_normals(v, t)
    cam = oracle.make_camera(c["K"], c["dist"], c["R"], c["t"], W, H)
    pt = oracle.project_points(cam, v)
    assert (np.abs(pt) >= 2.0 ** 31).any()                       # the case is present
    ok = (np.abs(pt) < 2.0 ** 31).all(1)
    r = np.where(ok[:, None], np.rint(np.where(ok[:, None], pt, 0)), -1).astype(np.int64)
    in_frame = ok & (r[:, 0] >= 0) & (r[:, 1] >= 0) & (r[:, 0] < W) & (r[:, 1] < H)
    res = oracle.create_projection(oracle.OracleBVH(s9), cam, v, nrm, tn,
                                   np.float32((180.0 - 70.0) * 3.141592653589793 / 180.0))
    # every in-frame node costs one primary ray; none of them needs a retry in this scene
    assert res["nrays"] == int(in_frame.sum()) == 35


def test_oblique_verdict_does_not_hang_on_the_last_bit_of_acos(oracle, fml):
    """psp_process.cpp:304-305 calls acos with a float argument in a TU that has <math.h>'s global overloads: acosf, whose
    last bit is libm's.  The oracle follows that overload; the GPU engine evaluates the double acos narrowed to float.  A
    node's entry is outside the bit-exact claim when the two land on opposite sides of deg2rad(180 - oblique): none on the
    reference's own grid + camera01, on the bench model (configs[1..3]) or on the config-1 sphere."""
    from upsp_processing_amd import synthetic as syn
    thr = np.float32((180.0 - 70.0) * 3.141592653589793 / 180.0)
    cam = oracle.make_camera(fml["cm"], fml["dist"].reshape(-1)[:4], fml["rmat"], fml["tvec"].reshape(3), 1024, 512)
    n_fml = oracle.oblique_ambiguous(cam, fml["nodes"], fml["norms"], thr)
    v, t = syn.tunnel_model_quad()
    c = syn.pinhole_camera(1024, 1024, center=(0, 0, 20), half_extent=6.0, fill=0.7)      # bench.py's camera
    n_bench = oracle.oblique_ambiguous(oracle.make_camera(c["K"], c["dist"], c["R"], c["t"], 1024, 1024), v,
                                       syn.node_normals(v, t), thr)
    v, t = syn.uv_sphere(50, 100)
    c = syn.pinhole_camera(512, 512)
    n_sph = oracle.oblique_ambiguous(oracle.make_camera(c["K"], c["dist"], c["R"], c["t"], 512, 512), v,
                                     syn.node_normals(v, t), thr)
    assert (n_fml, n_bench, n_sph) == (0, 0, 0)
    # the counter does see such nodes when they exist: normals turned so that cos(theta) sweeps every float around
    # cos(threshold) -- libm's acosf and the rounded double acos part company on some of them (or on none, where libm's
    # acosf is correctly rounded: then the two are the same function on these inputs)
    cs = np.float32(np.cos(np.float64(thr)))
    grid = cs + np.arange(-2000, 2000, dtype=np.float32) * np.float32(2.0 ** -25)
    a = np.arccos(grid.astype(np.float64)).astype(np.float32) > thr
    assert a.any() and (~a).any()                                # the sweep straddles the threshold
