"""Mirror of the reference's Python caller of the ray caster:
``upsp.cam_cal_utils.visibility.VisibilityChecker``
(python/upsp/cam_cal_utils/visibility.py).

Same names, argument meaning and results; the per-node Python loop over
``upsp.raycast.Ray / Hit / BVH.intersect`` (visibility.py:392-420, 464-488) is
replaced by one batched occlusion query on the GPU.  The float64 numpy
preparation (unit vectors, back-face rule, epsilon offset) is kept operation for
operation because it decides which rays are cast and where they start; rays are
narrowed to float32 at the binding exactly like the reference
(cpp/pybind11/raycast.cpp:17-22).
"""
import numpy as np


class VisibilityChecker:
    """Visibility of grid nodes from a camera: oblique-angle test + occlusion test.

    Parameters
    ----------
    scene : BVH
        object with ``occluded_many(origins, dirs)`` / ``intersect_many`` (the
        ``upsp_processing_amd.raycast`` module's BVH) -- the reference builds it with
        ``upsp.raycast.CreateBVH(primitives, 3)`` (visibility.py:130)
    oblique_angle : float
        maximum allowable oblique viewing angle in degrees (visibility.py:132-145)
    epsilon : float
        offset of the ray origin along the unit normal (visibility.py:117, 474-475)
    """

    def __init__(self, scene=None, oblique_angle=70, epsilon=1e-4, primitives=None):
        if scene is None:
            if primitives is None:
                raise ValueError("either a BVH or the triangle primitives are required")
            from . import raycast
            scene = raycast.CreateBVH(np.ascontiguousarray(primitives, dtype=np.float32), 3)
        self.scene = scene
        self.epsilon = epsilon
        self.update_oblique_angle(oblique_angle)

    def update_oblique_angle(self, oblique_angle):
        # visibility.py:132-145
        self.oblique_angle = oblique_angle
        self.squared_cos_angle = np.cos(np.deg2rad(oblique_angle)) ** 2

    def unit_vector(self, vector):
        # visibility.py:214-236
        return np.divide(vector, np.expand_dims(np.linalg.norm(vector, axis=1), 1))

    def angle_between(self, v1, v2):
        # visibility.py:238-264 (degrees)
        v1_u = self.unit_vector(v1)
        v2_u = self.unit_vector(v2)
        angle_r = np.arccos(np.clip(np.sum(np.multiply(v1_u, v2_u), axis=1), -1.0, 1.0))
        return np.rad2deg(angle_r)

    def is_back_facing_fast_vectorized(self, t, n):
        # visibility.py:362-390
        proj = np.sum(t * n, axis=-1)
        return np.where(proj * np.abs(proj) < self.squared_cos_angle * np.sum(t * t, axis=-1)
                        * np.sum(n * n, axis=-1), True, False)

    def does_intersect(self, origin, direction, return_pos=False):
        """visibility.py:392-420 for a single ray."""
        o = np.asarray(origin, dtype=np.float32).reshape(1, 3)
        d = np.asarray(direction, dtype=np.float32).reshape(1, 3)
        if return_pos:
            res = self.scene.intersect_many(o, d)
            return bool(res["hit"][0]), np.expand_dims(res["pos"][0], 1)
        return bool(self.scene.occluded_many(o, d)[0])

    def does_intersect_many(self, origins, directions):
        """Batched does_intersect: bool array, True where the ray hits the mesh."""
        o = np.ascontiguousarray(origins, dtype=np.float32).reshape(-1, 3)
        d = np.ascontiguousarray(directions, dtype=np.float32).reshape(-1, 3)
        if o.shape[0] == 0:
            return np.zeros(0, dtype=bool)
        return np.asarray(self.scene.occluded_many(o, d), dtype=bool)

    def is_visible(self, tvec_model_to_camera, nodes, normals, return_angles=False):
        """visibility.py:422-495: indices of nodes that are neither back facing nor occluded."""
        tvec_model_to_camera = np.asarray(tvec_model_to_camera, dtype=np.float64)
        nodes = np.asarray(nodes, dtype=np.float64)
        normals = np.asarray(normals, dtype=np.float64)
        tvecs = tvec_model_to_camera.T - nodes if tvec_model_to_camera.ndim == 2 \
            else tvec_model_to_camera - nodes
        tvec_norms = np.linalg.norm(tvecs, axis=-1)
        unit_tvecs = tvecs / np.expand_dims(tvec_norms, 1)
        normal_norms = np.linalg.norm(normals, axis=1)
        unit_normals = normals / np.expand_dims(normal_norms, 1)

        if not return_angles:
            back_facings = self.is_back_facing_fast_vectorized(unit_tvecs, unit_normals)
        else:
            angles = self.angle_between(unit_tvecs, unit_normals)
            back_facings = np.where(angles > self.oblique_angle, True, False)

        origins = nodes + self.epsilon * unit_normals
        cand = np.nonzero(~back_facings)[0]
        occluded = self.does_intersect_many(origins[cand], unit_tvecs[cand])
        visible = cand[~occluded]
        if return_angles:
            return np.array(visible), np.array(angles[visible])
        return np.array(visible, dtype=int)
