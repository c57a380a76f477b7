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


def inv_transform(R, t):
    """photogrammetry.invTransform (python/upsp/cam_cal_utils/photogrammetry.py:51-69)."""
    Rt = np.asarray(R, dtype=np.float64).transpose()
    return Rt, -np.matmul(Rt, np.asarray(t, dtype=np.float64))


def pts_inside_incal(rmat, tvec, cameraMatrix, distCoeffs, obj_pts, cal_area_is_safe=None,
                     cal_vol_is_safe=None, critical_pt=None, project=None):
    """internal_calibration.get_pts_inside_incal
    (python/upsp/cam_cal_utils/internal_calibration.py:16-300): indices of the points that lie in
    the well-behaved region of the lens model.

    Steps 1 / 2 (safe image area / safe volume) run only when the caller hands over the
    ``alpha_shape_is_safe`` callables (the reference's default stand-ins accept every point,
    internal_calibration.py:104-108); step 1 then needs ``project`` = callable(obj_pts) -> [n, 2]
    pixel positions (cv2.projectPoints of the reference, e.g. engine.project_points).  Step 3
    (``critical_pt`` 'first' / 'final'): the homogeneous coordinates must stay inside the extrema of
    the distortion polynomial; one numpy.roots call per point and axis like the reference (:191-292),
    so that the root selection (|imag| < 1e-10) sees the same eigenvalue noise."""
    assert critical_pt in ["first", "final", None]
    obj_pts = np.asarray(obj_pts, dtype=np.float64).reshape(-1, 3)
    distCoeffs = np.asarray(distCoeffs, dtype=np.float64).reshape(1, -1)
    rmat = np.asarray(rmat, dtype=np.float64)
    tvec = np.asarray(tvec, dtype=np.float64).reshape(3, 1)
    safe = np.full(len(obj_pts), True)
    rel = (rmat @ obj_pts.T + tvec).T                       # photogrammetry.transform_3d_point
    if cal_area_is_safe is not None:
        if project is None:
            raise ValueError("cal_area_is_safe needs `project`")
        safe *= np.asarray(cal_area_is_safe(project(obj_pts)), dtype=bool)
    if cal_vol_is_safe is not None:
        safe *= np.asarray(cal_vol_is_safe(rel), dtype=bool)
    if critical_pt is None or (distCoeffs >= 0).all():      # :124-131
        return np.argwhere(safe == True).flatten()          # noqa: E712
    k1, k2, p1, p2, k3 = distCoeffs[0][:5]
    highest = k3 if k3 != 0.0 else k2 if k2 != 0.0 else k1 if k1 != 0.0 else min(p2, p1)   # :143-152
    if critical_pt == "final" and highest > 0.0:
        return np.argwhere(safe == True).flatten()          # noqa: E712
    xh = (rel[:, 0] / rel[:, 2]).astype(complex)
    yh = (rel[:, 1] / rel[:, 2]).astype(complex)

    def bounds(a, b, pa, pb):
        """Extrema of the projection of coordinate `a` as a function of a, the other coordinate b
        held fixed (:177-226 for x, :236-292 for y with p1 / p2 swapped)."""
        lo = np.full(a.shape, -np.inf)
        hi = np.full(a.shape, np.inf)
        b2, b4, b6 = np.power(b, 2), np.power(b, 4), np.power(b, 6)
        for i in range(len(a)):
            if not safe[i]:
                continue
            coeffs = np.array([k3, 0, k2 + 3 * k3 * b2[i], 0, k1 + 2 * k2 * b2[i] + 3 * k3 * b4[i], 3 * pb,
                               1 + k1 * b2[i] + k2 * b4[i] + k3 * b6[i] + 2 * pa * b[i], pb * b2[i]])
            roots = np.roots(np.polyder(coeffs))
            real_roots = np.real(roots[np.where(np.abs(np.imag(roots)) < 1e-10)])
            for r in real_roots:
                if critical_pt == "first":
                    if r > 0.0:
                        hi[i] = np.minimum(hi[i], r)
                    else:
                        lo[i] = np.maximum(lo[i], r)
                else:
                    lo[i] = np.minimum(lo[i], r)
                    hi[i] = np.maximum(hi[i], r)
        return lo, hi

    lo, hi = bounds(xh, yh, p1, p2)
    safe *= np.where(xh > lo, True, False)
    safe *= np.where(xh < hi, True, False)
    lo, hi = bounds(yh, xh, p2, p1)
    safe *= np.where(yh > lo, True, False)
    safe *= np.where(yh < hi, True, False)
    return np.argwhere(safe == True).flatten()              # noqa: E712


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

    def is_visible_and_inside_incal(self, rmat, tvec, cameraMatrix, distCoeffs, nodes, normals,
                                    incal_inputs=None):
        """visibility.py:497-567: indices of the nodes that are visible AND inside the well-behaved
        region of the internal calibration (rmat / tvec: camera to object)."""
        if incal_inputs is None:
            incal_inputs = {"critical_pt": "first"}
        _, tvec_model_to_cam = inv_transform(rmat, tvec)
        vis = self.is_visible(tvec_model_to_cam, nodes, normals, return_angles=False)
        nodes = np.asarray(nodes, dtype=np.float64)
        inside = pts_inside_incal(rmat, tvec, cameraMatrix, distCoeffs, nodes[vis], **incal_inputs)
        return np.sort(vis[inside])

    def get_occlusions(self, rmat, tvec, tvecs, norms):
        """photogrammetry.get_occlusions_targets (photogrammetry.py:339-392) on arrays: for every
        point the (occluded?, hit position) of the ray from the point (offset by epsilon along its
        normal) towards the camera -- one batched closest-hit query."""
        _, cam = inv_transform(rmat, tvec)
        p = np.asarray(tvecs, dtype=np.float64).reshape(-1, 3)
        n = np.asarray(norms, dtype=np.float64).reshape(-1, 3)
        d = cam.reshape(1, 3) - p
        d /= np.linalg.norm(d, axis=1, keepdims=True)
        o = p + self.epsilon * n
        res = self.scene.intersect_many(np.ascontiguousarray(o, dtype=np.float32),
                                        np.ascontiguousarray(d, dtype=np.float32))
        hit = np.asarray(res["hit"], dtype=bool)
        pos = np.where(hit[:, None], np.asarray(res["pos"], dtype=np.float64), 0.0)
        return [(bool(h), np.expand_dims(q, 1)) for h, q in zip(hit, pos)]
