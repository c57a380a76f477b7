"""Known-answer vectors the reference's own Python tests hold for the ray-cast path, replayed on the
CPU oracle (GPU: tests/test_reference_pins_gpu.py):

* test/python/test_photogrammetry.py:274-292  project_3d_point -> 8 pixel positions (2 decimals):
  the only reference-held pin on cv::projectPoints (k1 = -0.091), which the oracle restates;
* test/python/test_photogrammetry.py:312-365  get_visible_targets + transform_targets: the first
  three visible targets in camera coordinates (3 decimals) -- visibility + lens-model step on targets;
* test/python/test_photogrammetry.py:401-428  get_occlusions_targets: the ONLY reference vector on
  Hit.pos / the closest hit: [-10.70026875, -2.03368831, -7.0] (6 decimals);
* test/python/test_visibility.py:149-171      is_visible_and_inside_incal -> 117 553 nodes;
* test/python/test_target_bumping.py:60-160   does_intersect(target, normal, return_pos=True) of the 24 targets of
  the fml grid, seven of them pushed 0.02 INTO the model along their normals: exactly those seven come back as real
  occlusions (distance to the hit >= the tgts / grid tolerance), at a distance of 0.02 (2 decimals); pushed 0.02
  OUTWARDS, none of the 24 is occluded beyond the tolerance (weak pins: 2 decimals).
Inputs: the reference's fixtures under tests/golden/ and the calibration constants of its setUpClass
(tests/refdata.py).  The expected numbers are the reference's."""
import copy
import os

import numpy as np
import pytest

import refdata
from test_oracle_kat import OracleScene
from upsp_processing_amd.visibility import VisibilityChecker, inv_transform

CAL = refdata.PHOTOGRAMMETRY_CAL
TGTS = os.path.join(refdata.GOLDEN, "fml_tc3_volume.tgts")


@pytest.fixture(scope="module")
def checker(oracle, fml):
    return VisibilityChecker(OracleScene(oracle, fml["prims"]), oblique_angle=70, epsilon=1e-4)


def visible_targets(chk, tgts):
    """photogrammetry.get_visible_targets (photogrammetry.py:395-447)."""
    tv = np.squeeze(np.array([t["tvec"] for t in tgts]), 2)
    nm = np.squeeze(np.array([t["norm"] for t in tgts]), 2)
    idx = chk.is_visible_and_inside_incal(CAL["rmat"], CAL["tvec"], CAL["cameraMatrix"], CAL["distCoeffs"], tv, nm)
    return [tgts[i] for i in idx]


def check_projection_pins(project):
    tgts = refdata.read_tgts(TGTS)
    pts = np.array([t["tvec"] for t in tgts]).reshape(-1, 3)
    projs = project(pts)
    assert projs.shape == (24, 2)
    for i, want in ((0, [934.159, 118.636]), (2, [830.756, 257.950]), (4, [768.440, 396.979]),
                    (7, [678.448, 542.530]), (11, [481.248, 445.495]), (17, [-128.84, 184.490]),
                    (23, [-564.553, 413.039]), (-1, [-564.553, 413.039])):
        np.testing.assert_array_almost_equal(projs[i], want, decimal=2)


def check_target_pins(chk):
    tgts = refdata.read_tgts(TGTS)
    vis = visible_targets(chk, tgts)
    # transform_targets (photogrammetry.py:240-278) of the first three visible targets
    tf = [(CAL["rmat"] @ t["tvec"] + CAL["tvec"]).ravel() for t in vis[:3]]
    np.testing.assert_array_almost_equal(tf[0], [5.49454348, -1.89638399, 18.78280763], decimal=3)
    np.testing.assert_array_almost_equal(tf[1], [5.30912805, 2.76660932, 18.73047579], decimal=3)
    np.testing.assert_array_almost_equal(tf[2], [4.04636575, 0.0159757297, 18.7346435], decimal=3)
    nrm0 = (CAL["rmat"] @ vis[0]["norm"]).ravel()
    np.testing.assert_array_almost_equal(nrm0, [0.01945551, -0.01044645, -0.99975615], decimal=3)
    # get_occlusions_targets: the visible targets, one lifted by 0.25, one with its normal flipped
    tg = copy.deepcopy(vis)
    f1 = copy.deepcopy(tg[0]); f1["tvec"][2] += 0.25; tg.append(f1)
    f2 = copy.deepcopy(tg[0]); f2["norm"] *= -1; tg.append(f2)
    occ = chk.get_occlusions(CAL["rmat"], CAL["tvec"], [t["tvec"] for t in tg], [t["norm"] for t in tg])
    assert all(not o[0] for o in occ[:-1])
    assert occ[-1][0]
    np.testing.assert_array_almost_equal(occ[-1][1], np.array([[-10.70026875], [-2.03368831], [-7.0]]))
    return len(vis)


def check_bumping_pins(chk):
    """target_bumping.get_bumping_occlusion / is_real_occlusion / tgts_get_internals (python/upsp/target_operations/
    target_bumping.py:15-147) restated on the checker's does_intersect(return_pos=True), with the inputs and
    expectations of test/python/test_target_bumping.py:60-160."""
    tgts = refdata.read_tgts(TGTS)
    internals = [1, 2, 3, 5, 8, 13, 21]

    def occlusion(t):
        tv = np.array(t["tvec"], dtype=np.float64)
        nm = np.array(t["norm"], dtype=np.float64)
        nm /= np.linalg.norm(nm)
        eps, chk.epsilon = chk.epsilon, 0          # (the reference clears epsilon for this query: any occlusion counts)
        try:
            hit, pos = chk.does_intersect(tv, nm, return_pos=True)
        finally:
            chk.epsilon = eps
        return (True, float(np.linalg.norm(tv - pos))) if hit else (False, -1.0)

    tol = float(np.linalg.norm([np.sqrt(3) * 1e-4, np.sqrt(3) * 1e-3]))      # tgts_tol 1e-4, grid_tol 1e-3
    # pushed inwards: tgts_get_internals returns exactly the seven, and the hit is 0.02 away
    pushed = copy.deepcopy(tgts)
    for i in internals:
        pushed[i]["tvec"] = pushed[i]["tvec"] - 0.02 * pushed[i]["norm"]
    real, dist = [], {}
    for i, t in enumerate(pushed):
        hit, d = occlusion(t)
        if hit and d >= tol:
            real.append(t["name"])
        dist[i] = (hit, d)
    assert real == [tgts[i]["name"] for i in internals], (real, dist)
    for i in internals:
        assert dist[i][0] and abs(dist[i][1] - 0.02) < 5e-3, (i, dist[i])        # assertAlmostEqual(.., 0.02, 2)
    # untouched and pushed outwards: nothing beyond the tolerance
    for sign in (0.0, 1.0):
        moved = copy.deepcopy(tgts)
        for i in internals:
            moved[i]["tvec"] = moved[i]["tvec"] + sign * 0.02 * moved[i]["norm"]
        for i, t in enumerate(moved):
            hit, d = occlusion(t)
            assert not (hit and d >= tol), (sign, i, d)
    return len(real)


def test_target_bumping_pins(checker):
    assert check_bumping_pins(checker) == 7


def test_project_3d_point_pins(oracle):
    cam = oracle.make_camera(CAL["cameraMatrix"], CAL["distCoeffs"].ravel(), CAL["rmat"], CAL["tvec"].ravel(), 1024, 512)
    check_projection_pins(lambda pts: oracle.project_points(cam, pts).astype(np.float64))


def test_visible_targets_and_hit_position(checker):
    assert check_target_pins(checker) > 3


def test_is_visible_and_inside_incal_117553(checker, fml):
    dc = copy.deepcopy(CAL["distCoeffs"])
    dc[0][0] *= -1
    dc[0][1] = -0.4
    got = checker.is_visible_and_inside_incal(CAL["rmat"], CAL["tvec"], CAL["cameraMatrix"], dc, fml["nodes"], fml["norms"],
                                              {"critical_pt": "first"})
    assert len(got) == 117553
    assert np.all(np.diff(got) > 0)
