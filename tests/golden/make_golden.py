#!/usr/bin/env python3
"""Generates / re-validates the committed fixtures under tests/golden/.

Runs ONLY in the build container (needs /root/reference).  It

1. copies EVERY reference-held data file the tests use (FIXTURES below: the fml grid, camera calibration,
   target list, tunnel-conditions sample, the PLOT3D / TriModel sample files; the MRAW pair is handled by
   make_golden_video.py), checks that the committed copies are byte-identical to the reference's, and records
   each one's origin in the manifest (`reference_files`, `reference_data_fixtures`) -- regeneration cannot
   silently drop provenance;
2. imports the reference's *Python* input-preparation code unmodified
   (upsp.processing.p3d_utilities / p3d_conversions, upsp.cam_cal_utils.parsers,
   VisibilityChecker.package_primitives / get_tvecs_and_norms) and checks that
   tests/refdata.py reproduces its outputs bit for bit on the fixture grid.  The compiled
   `upsp.raycast` extension and `cv2` do not exist in this image; empty module objects are
   registered under those names only so that `import` statements succeed -- no function of
   either is called (the ray caster under test is this repository's);
3. runs the CPU oracle through the VisibilityChecker mirror and checks the reference's pinned
   count (test/python/test_visibility.py:243-254: 148 608), then stores the visible index set
   (oracle output, used as regression vector for the GPU path) and a manifest with SHA-256s.
"""
import hashlib
import importlib.util
import json
import os
import shutil
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


# committed fixture (relative to tests/golden/) -> (path under /root/reference, what it is / which reference test holds its expectations)
FIXTURES = {
    "fml_tc3_volume.grid": ("test/data/fml_tc3_volume.grid",
                            "PLOT3D surface grid of the reference's regression model (test/python/test_visibility.py)"),
    "camera01_35_6.json": ("test/data/camera-tunnel-calibration/camera01_35_6.json",
                           "camera-to-tunnel calibration of camera 1"),
    "wtd_test.wtd": ("test/data/wtd_test.wtd", "tunnel-conditions sample read by read_tunnel_conditions"),
    "fml_tc3_volume.tgts": ("test/data/fml_tc3_volume.tgts", "target / fiducial list of the fml grid"),
}
for _n in ("sphere_unf_single_integration_sp.x", "sphere_unf_single_integration_dp.x",
           "sphere_unf_single_integration_sp_bigend.x", "sphere_unf_single_integration_dp_bigend.x",
           "sphere_unf_multi_integration_sp.x", "sphere_unf_multi_integration_dp.x",
           "sphere_unf_multi_integration_sp_bigend.x", "sphere_unf_multi_integration_dp_bigend.x",
           "sphere_unf_multi_integration_sp_iblank.x", "sphere_unf_multi_integration_dp_iblank.x",
           "sphere_unf_single.tri", "sphere_unf_multi.tri", "sphere_unf_multi.i.tri"):
    FIXTURES["p3d/" + _n] = ("cpp/test/inputs/" + _n,
                             "PLOT3D / TriModel sample (expectations: cpp/test/test_plot3d.cpp:5-128, cpp/test/test_trimodel.cpp:63-141)")
for _n in ("26-scalars-with-seps.f", "26-scalars-without-seps.f"):
    FIXTURES["p3d/" + _n] = ("cpp/test/sample_data/" + _n, "PLOT3D function-file sample (cpp/test/test_plot3d.cpp)")


def file_sha(path):
    return hashlib.sha256(open(path, "rb").read()).hexdigest()


def sync_fixtures():
    """Copy missing fixtures from the reference, verify the committed ones byte for byte; returns the provenance block."""
    prov = {}
    for name, (src, what) in sorted(FIXTURES.items()):
        dst, ref = os.path.join(HERE, name), os.path.join(REF, src)
        assert os.path.exists(ref), "reference fixture missing: " + src
        if not os.path.exists(dst):
            os.makedirs(os.path.dirname(dst), exist_ok=True)
            shutil.copyfile(ref, dst)
            os.chmod(dst, 0o644)
        assert file_sha(dst) == file_sha(ref), "committed fixture %s differs from %s" % (name, src)
        prov[name] = {"from": src, "what": what + "; data file, copied verbatim", "sha256": file_sha(dst)}
    return prov


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def main():
    assert os.path.isdir(REF), "reference tree not mounted"
    prov = sync_fixtures()
    if "--fixtures-only" in sys.argv:             # provenance block only (no reference Python imported)
        mpath = os.path.join(HERE, "golden_manifest.json")
        manifest = json.load(open(mpath))
        manifest["reference_files"] = sorted(v["from"] for v in prov.values())
        manifest["reference_data_fixtures"] = prov
        json.dump(manifest, open(mpath, "w"), indent=1)
        print("provenance of %d fixtures written" % len(prov))
        return
    if not hasattr(np, "product"):
        np.product = np.prod                      # removed in NumPy 2, used by p3d_utilities.py:114
    for name in ("cv2", "upsp.raycast"):          # import-only placeholders, never called
        sys.modules.setdefault(name, types.ModuleType(name))
    sys.path.insert(0, os.path.join(REF, "python"))
    sys.dont_write_bytecode = True
    import upsp.processing.p3d_utilities as p3d
    import upsp.processing.p3d_conversions as p2g
    from upsp.cam_cal_utils import parsers, visibility

    import refdata
    grid = os.path.join(HERE, "fml_tc3_volume.grid")
    grd = p3d.read_p3d_grid(grid)
    t = p2g.p3d_to_gltf_triangles(grd)
    verts_ref = np.array(t["vertices"]).reshape(-1, 3)
    inds_ref = np.array(t["indices"], dtype=int).reshape(-1, 3)
    verts, inds = refdata.fml_grid()
    assert np.array_equal(verts, verts_ref) and np.array_equal(inds, inds_ref)

    vc = visibility.VisibilityChecker.__new__(visibility.VisibilityChecker)
    vc.grid_path = grid
    prims_ref = vc.package_primitives({"vertices": verts_ref, "indices": inds_ref})
    prims = refdata.package_primitives(verts, inds)
    assert np.array_equal(prims, prims_ref)
    nodes_ref, norms_ref = vc.get_tvecs_and_norms()
    nodes, norms, faces, fn = refdata.tvecs_and_norms(verts, inds)
    assert np.array_equal(nodes, nodes_ref) and np.array_equal(norms, norms_ref)
    assert faces.shape == (609120, 3, 3) and nodes.shape == (304566, 3)

    cal = os.path.join(HERE, "camera01_35_6.json")
    rm_ref, tv_ref, cm_ref, dc_ref = parsers.read_camera_tunnel_cal(cal, (512, 1024))
    rm, tv, cm, dc = refdata.read_camera_tunnel_cal(cal, (512, 1024))
    assert all(np.array_equal(a, b) for a, b in ((rm, rm_ref), (tv, tv_ref), (cm, cm_ref), (dc, dc_ref)))

    # targets file reader (parsers.read_tgts) -- fixture test/data/fml_tc3_volume.tgts
    tg_ref = parsers.read_tgts(os.path.join(REF, "test/data/fml_tc3_volume.tgts"))
    tg = refdata.read_tgts(os.path.join(HERE, "fml_tc3_volume.tgts"))
    assert len(tg) == len(tg_ref) == 24
    for a, b in zip(tg, tg_ref):
        assert a.keys() == b.keys()
        for k in a:
            assert np.array_equal(a[k], b[k]) if isinstance(a[k], np.ndarray) else a[k] == b[k], k

    from oracle import oracle as orc
    from test_oracle_kat import OracleScene
    from upsp_processing_amd.visibility import VisibilityChecker
    mirror = VisibilityChecker(OracleScene(orc, prims.astype(np.float32)), oblique_angle=70, epsilon=1e-4)
    vis = mirror.is_visible(-(rm.T @ tv), nodes, norms)
    assert len(vis) == 148608, len(vis)
    np.savez_compressed(os.path.join(HERE, "camera01_visible.npz"), visible=vis.astype(np.int32))
    manifest = {
        "reference_files": sorted(v["from"] for v in prov.values()),
        "reference_data_fixtures": prov,
        "primitives_sha256": sha(prims.astype(np.float32)),
        "nodes_sha256": sha(nodes), "normals_sha256": sha(norms),
        "visible_count": int(len(vis)), "visible_sha256": sha(vis.astype(np.int32)),
        "reference_pin": "test/python/test_visibility.py:243-254 (148608)",
    }
    json.dump(manifest, open(os.path.join(HERE, "golden_manifest.json"), "w"), indent=1)
    print(json.dumps(manifest, indent=1))


if __name__ == "__main__":
    main()
