"""GPU: the reference-held known-answer vectors of tests/test_reference_pins.py through the product
(pybind `raycast` BVH on the GPU, VisibilityChecker mirror, upsp_project_points_host)."""
import copy

import numpy as np
import pytest

import refdata
from test_reference_pins import CAL, check_bumping_pins, check_projection_pins, check_target_pins
from upsp_processing_amd.visibility import VisibilityChecker

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def checker(gpu_lib, fml):
    return VisibilityChecker(primitives=fml["prims"], oblique_angle=70, epsilon=1e-4)


def test_project_3d_point_pins_gpu(gpu_lib):
    from upsp_processing_amd import _capi, engine
    cam = _capi.make_camera(CAL["cameraMatrix"], CAL["distCoeffs"].ravel()[:4], CAL["rmat"], CAL["tvec"].ravel(), 1024, 512)
    check_projection_pins(lambda pts: np.asarray(engine.project_points(cam, pts), dtype=np.float64))


def test_visible_targets_and_hit_position_gpu(checker):
    assert type(checker.scene).__name__ == "BVH"
    assert check_target_pins(checker) > 3


def test_is_visible_and_inside_incal_117553_gpu(checker, fml):
    dc = copy.deepcopy(CAL["distCoeffs"])
    dc[0][0] *= -1
    dc[0][1] = -0.4
    got = checker.is_visible_and_inside_incal(CAL["rmat"], CAL["tvec"], CAL["cameraMatrix"], dc, fml["nodes"], fml["norms"],
                                              {"critical_pt": "first"})
    assert len(got) == 117553


def test_target_bumping_pins_gpu(checker):
    """test/python/test_target_bumping.py:60-160 through the GPU closest-hit query (does_intersect(return_pos=True))."""
    assert check_bumping_pins(checker) == 7
