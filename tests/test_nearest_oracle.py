"""Pins the kd-tree restatement (oracle/kd_oracle.c) on the reference's own kd-tree
(cpp/raycast/pspKdtree.c compiled into oracle/_ref/ where /root/reference exists; the built
library travels with the snapshot, the sources do not)."""
import os

import numpy as np
import pytest


def _cloud(n, seed, dup=True):
    rng = np.random.default_rng(seed)
    p = rng.normal(size=(n, 3)).astype(np.float32) * np.array([8, 2, 2], np.float32)
    if dup:                       # coincident nodes (zone overlaps of a PLOT3D grid) and lattice ties
        p[n // 2:n // 2 + n // 10] = p[:n // 10]
        p[-50:] = np.round(p[-50:])
    return p


def test_restatement_basic(oracle):
    p = _cloud(3000, 1, dup=False)
    rng = np.random.default_rng(2)
    q = rng.normal(size=(200, 3)) * np.array([8, 2, 2])
    idx, d2 = oracle.KdTree(p).nearest(q)
    D = ((p.astype(np.float64)[None] - q[:, None]) ** 2)
    full = (D[..., 0] + D[..., 1]) + D[..., 2]
    assert np.array_equal(d2, full.min(1))
    assert np.array_equal(idx, full.argmin(1))          # no exact ties in a random cloud


def test_restatement_matches_reference_kdtree(oracle):
    if not oracle.build_ref():
        pytest.skip("oracle/_ref/libpspkdtree.so not built (no /root/reference on this machine)")
    for n, seed in ((1, 3), (2, 4), (500, 5), (4000, 6)):
        p = _cloud(n, seed, dup=n >= 500)
        rng = np.random.default_rng(seed + 100)
        q = np.concatenate([rng.normal(size=(150, 3)) * np.array([8, 2, 2]),
                            p[rng.integers(0, n, 60)].astype(np.float64),           # exact hits
                            np.round(rng.normal(size=(60, 3)) * 3) + 0.5])          # lattice ties
        ref = oracle.RefKdTree(p).nearest(q)
        got, _ = oracle.KdTree(p).nearest(q)
        assert np.array_equal(ref, got), (n, np.nonzero(ref != got)[0][:5])


def test_restatement_on_reference_grid(oracle, fml):
    """Nodes of test/data/fml_tc3_volume.grid (zone overlaps -> coincident nodes)."""
    if not oracle.build_ref():
        pytest.skip("oracle/_ref/libpspkdtree.so not built")
    nodes = fml["nodes"][::7].astype(np.float32)         # 44k nodes keep the ctypes loop short
    rng = np.random.default_rng(17)
    q = np.concatenate([nodes[rng.integers(0, len(nodes), 150)].astype(np.float64) + rng.normal(size=(150, 3)) * 0.05,
                        nodes[rng.integers(0, len(nodes), 50)].astype(np.float64)])
    ref = oracle.RefKdTree(nodes).nearest(q)
    got, _ = oracle.KdTree(nodes).nearest(q)
    assert np.array_equal(ref, got)
