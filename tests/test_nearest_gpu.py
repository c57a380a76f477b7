"""Nearest-node query on the GPU (exhaustive scan) vs the kd-tree oracle.
Bit-exact squared distances (same double arithmetic); same node unless two nodes are
exactly equidistant, where the scan returns the lowest index (documented in upsp_gpu.h)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n,nq", [(1, 3), (63, 5), (5000, 257), (300000, 40)])
def test_nearest_vs_kdtree(gpu_lib, oracle, n, nq):
    from upsp_processing_amd import engine
    rng = np.random.default_rng(n)
    p = (rng.normal(size=(n, 3)) * np.array([8, 2, 2])).astype(np.float32)
    q = rng.normal(size=(nq, 3)) * np.array([8, 2, 2])
    q[: nq // 3] = p[rng.integers(0, n, nq // 3)]                      # exact hits
    want, wd2 = oracle.KdTree(p).nearest(q)
    idx, d2 = engine.nearest_nodes(p, q, want_dist=True)
    assert np.array_equal(d2.cpu().numpy(), wd2)
    assert np.array_equal(idx.cpu().numpy(), want)


def test_nearest_ties_lowest_index(gpu_lib, oracle):
    from upsp_processing_amd import engine
    rng = np.random.default_rng(3)
    p = (rng.normal(size=(2000, 3)) * 4).astype(np.float32)
    p[1000:1200] = p[:200]                                              # coincident nodes
    q = p[rng.integers(0, 1200, 300)].astype(np.float64)
    want, wd2 = oracle.KdTree(p).nearest(q)
    idx, d2 = engine.nearest_nodes(p, q, want_dist=True)
    idx = idx.cpu().numpy()
    assert np.array_equal(d2.cpu().numpy(), wd2)                        # the same distance, always
    same_pos = np.all(p[idx] == p[want], axis=1)
    assert same_pos.all() and np.all(idx <= want)                       # a coincident node, lowest index


def test_nearest_on_reference_grid(gpu_lib, oracle, fml):
    from upsp_processing_amd import engine
    nodes = fml["nodes"].astype(np.float32)
    rng = np.random.default_rng(23)
    q = nodes[rng.integers(0, len(nodes), 200)].astype(np.float64) + rng.normal(size=(200, 3)) * 0.05
    want, wd2 = oracle.KdTree(nodes).nearest(q)
    idx, d2 = engine.nearest_nodes(nodes, q, want_dist=True)
    assert np.array_equal(d2.cpu().numpy(), wd2)
    idx = idx.cpu().numpy()
    diff = idx != want
    assert np.all(np.all(nodes[idx[diff]] == nodes[want[diff]], axis=1))   # only coincident nodes differ


def test_nearest_errors(gpu_lib):
    import torch
    from upsp_processing_amd import engine
    with pytest.raises(Exception):
        engine.nearest_nodes(torch.zeros((0, 3)), np.zeros((2, 3)))
    assert engine.nearest_nodes(np.zeros((4, 3), np.float32), np.zeros((0, 3))).numel() == 0
