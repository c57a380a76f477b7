"""Host logic of the phase-0 patch set-up (upsp_processing_amd/patch_setup.py) against the C
oracle (oracle/patchsetup_oracle.c): two independent restatements of
cpp/lib/patches.ipp / cpp/utils/clustering.ipp / cpp/lib/image_processing.ipp.
Integer / index work: bit-exact."""
import os

import numpy as np
import pytest

from upsp_processing_amd import patch_setup as ps

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_read_target_file():
    t = ps.read_psp_target_file(os.path.join(GOLD, "fml_tc3_volume.tgts"))
    assert len(t) == 24 and t[0].num == 1 and t[23].num == 24
    assert np.allclose(t[0].xyz, [-10.7003, -2.0337, -7.0]) and abs(float(t[0].diameter) - 0.063) < 1e-7
    taps = ps.read_psp_target_file(os.path.join(GOLD, "fml_tc3_volume.tgts"), "*Taps")
    assert len(taps) > 0 and abs(float(taps[0].diameter) - 0.020) < 1e-7
    assert ps.read_psp_target_file(os.path.join(GOLD, "fml_tc3_volume.tgts"), "*Fiducials") == []


def test_read_target_file_short_lines(tmp_path):
    p = tmp_path / "t.tgts"
    p.write_text("#x\n*Targets\n 1 1.0 2.0 3.0 0 0 1 0.5 a b\n\n 3 4.0\n*Taps\n 9 9 9 9 0 0 1 0.1\n")
    t = ps.read_psp_target_file(str(p))
    assert [x.num for x in t] == [1, 0, 3]
    assert np.allclose(t[1].xyz, [1, 2, 3]) and float(t[1].diameter) == 0.5      # blank line: previous values
    assert np.allclose(t[2].xyz, [4, 0, 3]) and float(t[2].diameter) == 0.5      # first failed field zeroed
    assert np.allclose(ps.read_psp_target_file(str(p), planar=True)[0].xyz, [1, 2, 0])


def _targets(n, seed, spread=60.0):
    rng = np.random.default_rng(seed)
    uv = (rng.random((n, 2)) * spread + 20).astype(np.float32)
    diam = (rng.random(n) * 6 + 2).astype(np.float32)
    diam[rng.random(n) < 0.1] = 0
    return [ps.Target((0, 0, 0), uv[i], diam[i], i) for i in range(n)], uv, diam


@pytest.mark.parametrize("n,seed,bp", [(1, 0, 3), (12, 1, 3), (40, 2, 3), (40, 3, 8), (25, 4, 0)])
def test_cluster_points(oracle, n, seed, bp):
    targs, uv, diam = _targets(n, seed)
    cl = ps.cluster_points(targs, bp)
    order, off = oracle.cluster_points(uv, diam, bp)
    assert [len(c) for c in cl] == np.diff(off).tolist()
    assert [t.num for c in cl for t in c] == order.tolist()


@pytest.mark.parametrize("seed,bp,buf", [(5, 2, 1), (6, 2, 0), (7, 0, 1), (8, 3, 2), (9, 1, 1)])
def test_patch_tables(oracle, seed, bp, buf):
    targs, uv, diam = _targets(30, seed, spread=90.0)
    size = (128, 100)
    uv[0] = (1.5, 2.5)                  # patches that leave the frame
    uv[1] = (126.0, 98.0)
    targs[0].uv, targs[1].uv = uv[0], uv[1]
    cl = ps.cluster_points(targs, bp + buf)
    order, off = oracle.cluster_points(uv, diam, bp + buf)
    got = ps.patch_clusters(cl, size, bp, buf)
    want = oracle.patch_tables(uv, diam, order, off, size, bp, buf)
    assert len(got) == len(want) and any(len(c) > 1 for c in cl) and any(len(c) == 1 for c in cl)
    for g, w in zip(got, want):
        for k in ("ix", "iy", "bx", "by"):
            assert np.array_equal(g[k], w[k]), k
    # threshold_bounds on a frame with dark discs
    rng = np.random.default_rng(seed)
    ref = (1500 + 100 * rng.standard_normal((size[1], size[0]))).clip(0, 4095).astype(np.uint16)
    yy, xx = np.mgrid[:size[1], :size[0]]
    for t in targs[:15]:
        ref[(xx - t.uv[0]) ** 2 + (yy - t.uv[1]) ** 2 < (0.5 * float(t.diameter) + 2.5) ** 2] = 300
    got = ps.threshold_bounds(got, ref, 900, 2)
    want = oracle.patch_tables(uv, diam, order, off, size, bp, buf, ref=ref, thresh=900, offset=2)
    nb = 0
    for g, w in zip(got, want):
        assert np.array_equal(g["bx"], w["bx"]) and np.array_equal(g["by"], w["by"])
        nb += g["bx"].size
    assert nb > 0 or bp == 0


def test_single_target_counts():
    t = ps.Target((0, 0, 0), (50.25, 40.75), 4.0, 1)
    internal, bounds = ps.get_target_boundary(t, 2, 1)
    # box: x 48..53, y 38..43 -> 36 interior; frame of thickness 2 at distance 1: 12x12 - 8x8
    assert len(internal) == 36 and len(bounds) == 12 * 12 - 8 * 8
    assert internal[0] == (48, 38) and internal[1] == (48, 39)          # x outer, y inner


def test_histogram_and_threshold(oracle):
    rng = np.random.default_rng(3)
    img = np.concatenate([rng.normal(300, 40, 20000), rng.normal(1800, 200, 80000)]).clip(0, 5000)
    img = img.astype(np.uint16).reshape(250, 400)
    e, c = ps.intensity_histc(img, 12, 256)
    eo, co = oracle.intensity_histc(img, 12, 256)
    assert np.array_equal(e, eo) and np.array_equal(c, co) and c.sum() == (img < 4096).sum()
    assert ps.first_min_threshold(c, 5) == oracle.first_min_threshold(c, 5)
    e2, c2 = ps.intensity_histc(img, 12, -1)
    eo2, co2 = oracle.intensity_histc(img, 12, -1)
    assert np.array_equal(e2, eo2) and np.array_equal(c2, co2)


@pytest.mark.parametrize("seed", range(8))
def test_find_peaks_and_first_min(oracle, seed):
    rng = np.random.default_rng(seed)
    n = int(rng.integers(3, 80))
    counts = rng.integers(0, 6, n)                       # many plateaus, zeros -> inf in 1/counts
    if seed % 2:
        counts = np.repeat(counts, 2)[:n]
    for sep in (0, 1, 5):
        assert ps.find_peaks(counts.tolist(), sep) == oracle.find_peaks(counts, sep)
        with np.errstate(divide="ignore"):
            inv = 1.0 / counts
        assert ps.find_peaks(inv.tolist(), sep) == oracle.find_peaks(inv, sep)
        assert ps.first_min_threshold(counts, sep) == oracle.first_min_threshold(counts, sep)


def test_find_peaks_break_quirk():
    # a higher peak within the separation band replaces the previous one and ends the scan
    data = [0, 5, 0, 9, 0, 0, 0, 0, 0, 7, 0]
    assert ps.find_peaks(data, 0) == [1, 3, 9]
    assert ps.find_peaks(data, 5) == [3]


def test_get_perpendicular():
    for v in ([0, 0, 2], [1, 2, 3], [-3, 0.5, 0.1], [0.2, -5, 1]):
        p = ps.get_perpendicular(np.array(v, np.float32))
        assert abs(float(np.dot(p, np.array(v, np.float32)))) < 1e-5 and abs(np.linalg.norm(p) - 1) < 1e-6
    assert np.all(ps.get_perpendicular(np.zeros(3, np.float32)) == 0)
