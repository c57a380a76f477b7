"""upsp::interpolate (k-nearest inverse-distance weighting) on the GPU vs the exhaustive oracle.
Neighbour sets: identical (ties broken by index on both sides); values: bit-exact (same float
operations in the same neighbour order)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _surface(n, seed, spread=(8, 2, 0.3)):
    rng = np.random.default_rng(seed)
    u = rng.random((n, 2))
    return np.stack([(u[:, 0] - 0.5) * 2 * spread[0], (u[:, 1] - 0.5) * 2 * spread[1],
                     spread[2] * np.cos(3 * u[:, 0]) * np.sin(2 * u[:, 1])], axis=1).astype(np.float32)


@pytest.mark.parametrize("ns,nq,k", [(5, 40, 10), (300, 500, 10), (20000, 3000, 10), (4000, 1000, 3), (4000, 500, 16)])
def test_idw_vs_oracle(gpu_lib, oracle, ns, nq, k):
    from upsp_processing_amd import engine
    src = _surface(ns, ns)
    data = np.sin(src[:, 0]) + 0.3 * src[:, 1]
    qry = _surface(nq, nq + 1, spread=(8.5, 2.2, 0.35))          # some queries outside the source box
    qry[:10] = src[:10] if ns >= 10 else qry[:10]                # exact hits
    want, wn = oracle.interpolate_idw(src, data, qry, k, 2.0)
    got, gn = engine.interpolate_idw(src, data, qry, k, 2.0, want_neighbors=True)
    assert np.array_equal(gn.cpu().numpy(), wn)
    assert np.array_equal(got.cpu().numpy().view(np.int32), want.view(np.int32))
    if ns >= 10:
        assert np.array_equal(got.cpu().numpy()[:10], data[:10].astype(np.float32))


def test_idw_structured_to_unstructured(gpu_lib, oracle):
    """Steady-state Cp from a structured grid onto an unstructured model (psp_process.cpp:2374-2377)."""
    from upsp_processing_amd import engine, synthetic as syn
    v, t = syn.tunnel_model_quad(24, 8)                           # the model (unstructured view)
    J, K = 120, 60                                                # a finer structured "steady" grid around it
    th, ph = np.meshgrid(np.linspace(0, 2 * np.pi, J), np.linspace(0.05, np.pi - 0.05, K))
    sg = np.stack([6 * np.cos(ph), np.sin(ph) * np.cos(th), np.sin(ph) * np.sin(th)], axis=-1).reshape(-1, 3).astype(np.float32)
    cp = (0.5 * np.cos(ph) ** 2 - 0.2).reshape(-1).astype(np.float32)
    want, _ = oracle.interpolate_idw(sg, cp, v, 10, 2.0)
    got = engine.interpolate_idw(sg, cp, v, 10, 2.0).cpu().numpy()
    assert np.array_equal(got.view(np.int32), want.view(np.int32))
    assert np.isfinite(got).all() and got.min() >= cp.min() - 1e-6 and got.max() <= cp.max() + 1e-6


def test_idw_errors(gpu_lib):
    from upsp_processing_amd import engine
    with pytest.raises(Exception):
        engine.interpolate_idw(np.zeros((4, 3)), np.zeros(4), np.zeros((2, 3)), k=17)
    with pytest.raises(ValueError):
        engine.interpolate_idw(np.zeros((4, 3)), np.zeros(3), np.zeros((2, 3)))
