"""Grid loaders and the structured-surface model, pinned on the reference's own tests:
cpp/test/test_plot3d.cpp (read/write round trips over endianness / precision / IBLANK, function
files), cpp/test/test_trimodel.cpp (Cart3D .tri sizes), cpp/test/test_p3dmodel.cpp with the
grid of cpp/test/test_grid_utils.cpp:49-122 (sizes, exact / toleranced overlap, normals)."""
import os

import numpy as np
import pytest

from upsp_processing_amd import grids, psp_process as cli

P3D = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "p3d")


def _same(a, b):
    return open(a, "rb").read() == open(b, "rb").read()


@pytest.mark.parametrize("kind", ["single", "multi"])
def test_plot3d_read_write_roundtrip(tmp_path, kind):
    """test_plot3d.cpp:34-128: every variant re-written equals the little-endian file of the
    precision it was read into (float reader <-> _sp, double reader <-> _dp)."""
    base = os.path.join(P3D, "sphere_unf_%s_integration_" % kind)
    out = str(tmp_path / "o.x")
    variants = ["sp", "sp_bigend", "dp", "dp_bigend"] + (["sp_iblank", "dp_iblank"] if kind == "multi" else [])
    for v in variants:
        for dtype, ref in ((np.float32, "sp"), (np.float64, "dp")):
            if v.endswith("bigend") or v.endswith("iblank"):
                if not v.startswith(ref):
                    continue                      # the reference cross-converts the plain files only
            g = grids.read_plot3d_grid(base + v + ".x", dtype)
            grids.write_plot3d_grid(out, g)
            assert _same(out, base + ref + ".x"), (v, ref)
    g = grids.read_plot3d_grid(base + "sp.x")
    assert len(g["zones"]) == (1 if kind == "single" else len(g["zones"])) and g["x"].dtype == np.float32


def test_plot3d_function_file():
    """test_plot3d.cpp:5-31"""
    w, wo = os.path.join(P3D, "26-scalars-with-seps.f"), os.path.join(P3D, "26-scalars-without-seps.f")
    for mode in (-1, 1):
        assert grids.read_plot3d_scalar_function_file(w, mode).size == 26
    for mode in (-1, 0):
        assert grids.read_plot3d_scalar_function_file(wo, mode).size == 26
    with pytest.raises(ValueError):
        grids.read_plot3d_scalar_function_file(w, 0)
    with pytest.raises(ValueError):
        grids.read_plot3d_scalar_function_file(wo, 1)
    a, b = grids.read_plot3d_scalar_function_file(w), grids.read_plot3d_scalar_function_file(wo)
    assert np.array_equal(b, np.arange(26, dtype=np.float32))
    # reference behaviour kept: with separators the data record is read WITHOUT skipping its
    # leading marker (plot3d.cpp:69), so the marker (104 as float bits) becomes the first scalar
    assert a.view(np.int32)[0] == 104 and np.array_equal(a[1:], b[:-1])


@pytest.mark.parametrize("name,nodes,comps", [("sphere_unf_single.tri", 594, 1), ("sphere_unf_multi.tri", 594, 6),
                                              ("sphere_unf_multi.i.tri", 514, 6)])
def test_tri_reader_reference_samples(name, nodes, comps):
    """test_trimodel.cpp:63-141 (no intersection pass): sizes and component counts."""
    xyz, tris, c = cli.read_tri_grid(os.path.join(P3D, name))
    assert xyz.shape == (nodes, 3) and tris.shape == (1024, 3)
    assert tris.min() == 0 and tris.max() == nodes - 1
    assert c is not None and np.unique(c).size == comps


def _kat_grid(offset=0.0):
    """create_single_structgrid + add_zones_structgrid (test_grid_utils.cpp:49-122)."""
    x, y = np.zeros(52, np.float32), np.zeros(52, np.float32)
    for k in range(5):
        for j in range(4):
            x[k * 4 + j], y[k * 4 + j] = j, k
    for k in range(4):
        for j in range(3):
            x[20 + k * 3 + j] = np.float32(j + 3) + np.float32(offset)
            y[20 + k * 3 + j] = k
    for k in range(4):
        for j in range(5):
            x[32 + k * 5 + j] = np.float32(6.0 - k + np.float32(offset))
            y[32 + k * 5 + j] = np.float32(j - 4.0 - np.float32(offset))
    return dict(zones=[(4, 5, 1), (3, 4, 1), (5, 4, 1)], x=x, y=y, z=np.zeros(52, np.float32))


def test_p3dmodel_sizes_and_normals():
    """test_p3dmodel.cpp:28-64, 103-108"""
    g = _kat_grid()
    single = grids.P3DModel(dict(zones=g["zones"][:1], x=g["x"][:20], y=g["y"][:20], z=g["z"][:20]), 1e-10)
    assert single.size() == 20 and single.number_of_faces() == 12
    m = grids.P3DModel(g, 1e-10)
    assert m.size() == 52 and m.number_of_faces() == 12 + 6 + 12
    assert list(m.start[:3]) == [0, 20, 32]
    assert np.array_equal(single.normals[0], [0, 0, 1])
    n = m.normals
    assert np.allclose(np.abs(n[:, 2]), 1) and np.allclose(n[:, :2], 0)


def test_p3dmodel_exact_overlap():
    """test_p3dmodel.cpp:192-204"""
    m = grids.P3DModel(_kat_grid(), 1e-10)
    sup = [n for n in range(52) if m.is_superceded(n)]
    assert sup == [20, 23, 26, 29, 41, 46, 51]
    src = m.overlap_source()
    assert src[20] == 3 and src[23] == 7 and src[29] == 15 and np.all(src[[0, 1, 2, 33]] == [0, 1, 2, 33])
    sol = np.arange(52, dtype=np.float32)
    adj = m.adjust_solution(sol)
    assert adj[20] == 3 and adj[3] == 3 and adj[51] == src[51]


def test_p3dmodel_tolerance_overlap():
    """test_p3dmodel.cpp:207-232: offset 0.1 -> no overlap with tol 0.09, seven with 0.100001"""
    assert not grids.P3DModel(_kat_grid(0.1), 0.09).overlap
    m = grids.P3DModel(_kat_grid(0.1), 0.100001)
    assert sum(m.is_superceded(n) for n in range(52)) == 7


def test_p3dmodel_triangles():
    m = grids.P3DModel(_kat_grid(), 1e-10)
    soup, tn = m.extract_tris()
    assert tn.size == 3 * 2 * 30 and soup.size == 9 * 2 * 30
    assert list(tn[:6]) == [0, 1, 5, 5, 4, 0]                 # (i0,i1,i2), (i2,i3,i0), P3DModel.ipp:289-312
    assert np.array_equal(soup.reshape(-1, 3), m.nodes()[tn])


def test_p3dmodel_reference_grid():
    """fml_tc3_volume.grid: counts pinned by test/python/test_visibility.py:227-241
    (609 120 triangles = 2 per quad)."""
    m = grids.P3DModel.from_file(os.path.join(os.path.dirname(P3D), "fml_tc3_volume.grid"), 1e-3)
    assert 2 * m.number_of_faces() == 609120
    soup, tn = m.extract_tris()
    assert tn.size == 3 * 609120
    assert len(m.overlap) > 0
    mag = np.linalg.norm(m.normals.astype(np.float64), axis=1)
    assert np.all((np.abs(mag - 1) < 1e-5) | (mag == 0))
    src = m.overlap_source()
    assert np.all(src <= np.arange(m.size())) and (src != np.arange(m.size())).sum() > 0
