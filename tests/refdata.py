"""Helpers that restate the *input preparation* the reference's Python tests perform
before they reach the ray caster (test/python/test_visibility.py setUpClass):

* read_p3d_grid            python/upsp/processing/p3d_utilities.py:87-139
* p3d_to_triangles         python/upsp/processing/p3d_conversions.py:201-222
* package_primitives       python/upsp/cam_cal_utils/visibility.py:167-212
* tvecs_and_norms          python/upsp/cam_cal_utils/visibility.py:591-657
* read_camera_tunnel_cal   python/upsp/cam_cal_utils/parsers.py:353-397

tests/golden/make_golden.py checks these against the reference's own modules
(in the build container, where /root/reference exists).  The data files under
tests/golden/ (fml_tc3_volume.grid, camera01_35_6.json) are fixtures held by the
reference's test suite (test/data/).
"""
import json
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def read_p3d_grid(filename):
    with open(filename, "rb") as f:
        np.fromfile(f, dtype=np.int32, count=1)
        n_zones = int(np.fromfile(f, dtype=np.int32, count=1)[0])
        np.fromfile(f, dtype=np.int32, count=1)
        np.fromfile(f, dtype=np.int32, count=1)
        zone_sz = np.fromfile(f, dtype=np.int32, count=n_zones * 3).reshape(n_zones, 3)
        np.fromfile(f, dtype=np.int32, count=1)
        xs, ys, zs = [], [], []
        for i in range(n_zones):
            zs_ = int(np.prod(zone_sz[i]))
            np.fromfile(f, dtype=np.int32, count=1)
            xyz = np.fromfile(f, dtype=np.float32, count=3 * zs_)
            np.fromfile(f, dtype=np.int32, count=1)
            xs.append(xyz[:zs_].astype(np.float64))
            ys.append(xyz[zs_:2 * zs_].astype(np.float64))
            zs.append(xyz[2 * zs_:].astype(np.float64))
    return zone_sz, np.concatenate(xs), np.concatenate(ys), np.concatenate(zs)


def p3d_to_triangles(zone_sz, x, y, z):
    """Returns (vertices [V,3] f64, indices [F,3] int) like p3d_to_gltf_triangles."""
    verts, inds = [], []
    idx0 = 0   # offset into the grid arrays
    for imax, jmax, kmax in zone_sz:
        imax, jmax = int(imax), int(jmax)
        n = imax * jmax
        verts.append(np.stack([x[idx0:idx0 + n], y[idx0:idx0 + n], z[idx0:idx0 + n]], axis=1))
        ii, jj = np.meshgrid(np.arange(imax - 1), np.arange(jmax - 1), indexing="ij")
        ii, jj = ii.ravel(), jj.ravel()
        p0 = idx0 + jj * imax + ii
        p1 = p0 + 1
        p2 = idx0 + (jj + 1) * imax + ii + 1
        p3 = idx0 + (jj + 1) * imax + ii
        quad = np.stack([p0, p1, p2, p0, p2, p3], axis=1).reshape(-1, 3)
        inds.append(quad)
        idx0 += n
    return np.concatenate(verts), np.concatenate(inds)


def _nondegenerate(faces):
    a, b, c = faces[:, 0], faces[:, 1], faces[:, 2]
    deg = (a == b).all(1) | (a == c).all(1) | (b == c).all(1)
    return ~deg


def package_primitives(verts, inds):
    """[T*9] float64 triangle soup, degenerate faces dropped."""
    faces = verts[inds]
    return faces[_nondegenerate(faces)].reshape(-1)


def tvecs_and_norms(verts, inds):
    """Unique nodes in first-appearance order and the normal of the first face
    that contains each of them."""
    faces = verts[inds]
    faces = faces[_nondegenerate(faces)]
    A = faces[:, 1] - faces[:, 0]
    B = faces[:, 2] - faces[:, 1]
    fn = np.cross(A, B)
    flat = faces.reshape(-1, 3) + 0.0          # -0.0 -> +0.0 so that equal tuples compare equal
    _, first = np.unique(flat, axis=0, return_index=True)
    first.sort()
    return flat[first], fn[first // 3], faces, fn


def read_camera_tunnel_cal(path, dims):
    """dims = (image height, image width).  Returns rmat, tvec(3,1), cameraMatrix, distCoeffs."""
    with open(path) as f:
        cal = json.load(f)
    upsp_cm = np.array(cal["uPSP_cameraMatrix"], dtype=np.float64)
    cm = upsp_cm.copy()
    # convert_uPSP_cm_to_cv2_cm (parsers.py): principal point is stored relative to image centre
    cm[0, 2] = upsp_cm[0, 2] + dims[1] / 2
    cm[1, 2] = upsp_cm[1, 2] + dims[0] / 2
    return (np.array(cal["rmat"]), np.array(cal["tvec"]).reshape(3, 1), cm,
            np.array(cal["distCoeffs"]))


def read_tgts(path, output_target_types=None):
    """parsers.read_tgts (python/upsp/cam_cal_utils/parsers.py:13-97): the *Targets section."""
    targets, section = [], None
    for raw in open(path):
        line = [x for x in raw.rstrip("\n").split(" ") if x != ""]
        if len(line) <= 1:                 # a one-item line names the section, an empty line ends it
            section = line[0] if len(line) == 1 else None
            continue
        if section != "*Targets":
            continue
        last = line[-1]
        ttype = "dot" if "st" in last else "kulite" if "mK" in last else "painted_kulite" if "pK" in last else last
        if output_target_types is None or ttype in output_target_types:
            targets.append({"target_type": ttype,
                            "tvec": np.expand_dims([float(x) for x in line[1:4]], 1),
                            "norm": np.expand_dims([float(x) for x in line[4:7]], 1),
                            "size": float(line[7]), "name": last, "idx": int(line[0]),
                            "zones": (int(line[8]), int(line[9]), int(line[10]))})
    return targets


# inputs of test/python/test_photogrammetry.py setUpClass (:24-34) -- reference-held test data
PHOTOGRAMMETRY_CAL = dict(
    rmat=np.array([[-0.999726480569, -0.0129787134506, 0.0194555145360],
                   [-0.013183724300, 0.9998585205361, -0.0104464503478],
                   [-0.019317180494, -0.0107000891804, -0.9997561475826]]),
    tvec=np.array([[-5.093035986816], [-0.07716666965650], [11.556054197934]]),
    cameraMatrix=np.array([[1380.2632820187425, 0.0, 533.908701486902032],
                           [0.0, 1380.2632820187425, 256.778541140320840],
                           [0.0, 0.0, 1.0]]),
    distCoeffs=np.array([[-0.09098491035825468, 0.0, 0.0, 0.0, 0.0]]))


def fml_grid():
    zs, x, y, z = read_p3d_grid(os.path.join(GOLDEN, "fml_tc3_volume.grid"))
    return p3d_to_triangles(zs, x, y, z)


def distorted_plates_scene():
    """Two axis-aligned plates seen by a camera with k1 = -0.05 whose far-off nodes project to
    |pt| ~ 1e10 .. 1e16 pixels (scene 6438 of tests/debug/soak_raycast.py, which exposed an int
    wrap-around in the oracle's cvRound).  Returns (verts, tris, camera dict, (W, H))."""
    import numpy as np
    from upsp_processing_amd import synthetic as syn
    rng = np.random.default_rng(6438)
    g = int(rng.integers(2, 30))
    x, y = np.meshgrid(np.arange(g + 1, dtype=np.float32), np.arange(g + 1, dtype=np.float32))
    v = np.stack([x.ravel(), y.ravel(), np.zeros(x.size, np.float32)], 1)
    q = np.arange(g * g)
    i0 = q // g * (g + 1) + q % g
    t = np.concatenate([np.stack([i0, i0 + 1, i0 + g + 2], 1),
                        np.stack([i0 + g + 2, i0 + g + 1, i0], 1)]).astype(np.int32)
    v2 = v.copy()
    v2[:, 2] = rng.choice([0.5, 1.0, 3.0])
    v = np.ascontiguousarray(np.concatenate([v, v2]), np.float32)
    t = np.concatenate([t, t + x.size])
    m = 4000                      # the soak draws its test rays here: keep the generator in step
    rng.normal(size=(m, 3)); rng.normal(size=(m, 3)); rng.integers(0, v.shape[0], m); rng.normal(size=(1, 3))
    rng.integers(0, 3, m); rng.choice([-1.0, 1.0, 2.5], m); rng.choice([0.0, 0.0, 0.1], (m, 3))
    W, H = int(rng.choice([64, 200, 512])), int(rng.choice([48, 160, 512]))
    cam = syn.pinhole_camera(W, H, center=tuple(rng.normal(size=3) * 3 + np.array([0, 0, 12])),
                             half_extent=float(rng.uniform(2, 7)), k1=float(rng.choice([0.0, -0.05])),
                             azimuth_deg=float(rng.uniform(0, 360)))
    return v, t, cam, (W, H)
