"""Oracle pinning for phase 2: the reference's TransPolyfitter known-answer test
(cpp/test/test_filtering.cpp:19-113) replayed on the oracle, plus the gain formula."""
import numpy as np


def _kat_inputs():
    # test_filtering.cpp:24-55 : degree 6, 25 frames, 13 points, coefficients 2.5/(c+1) + p/(c+1)
    degree, n_frames, n_pts = 6, 25, 13
    coeffs = degree + 1
    x = (np.arange(n_frames, dtype=np.float32) / np.float32(n_frames)).astype(np.float32)
    true = np.zeros((n_pts, coeffs), np.float32)
    for c in range(coeffs):
        for p in range(n_pts):
            true[p, c] = np.float32(2.5 / (c + 1) + p / (c + 1))
    y = np.zeros((n_pts, n_frames), np.float32)
    for p in range(n_pts):
        for f in range(n_frames):
            for c in range(coeffs):
                y[p, f] = np.float32(y[p, f] + np.float32(np.float32(float(x[f]) ** c) * true[p, c]))
    return degree, n_frames, n_pts, y


def test_transpolyfitter_kat(oracle):
    degree, n_frames, n_pts, y = _kat_inputs()
    eps = 1e-4                                      # test_filtering.cpp:22
    for p in range(n_pts):
        poly, fit = oracle.transpoly_fit(y[p], degree)
        assert np.max(np.abs(fit - y[p])) < eps


def test_design_matrix(oracle):
    A = oracle.transpoly_design(25, 6)
    x = np.arange(25, dtype=np.float32) / np.float32(25)
    want = np.stack([(x.astype(np.float64) ** c).astype(np.float32) for c in range(7)], axis=1)
    assert np.array_equal(A, want)


def test_fit_is_least_squares(oracle):
    rng = np.random.default_rng(5)
    F = 400
    y = (1.0 + 0.01 * rng.standard_normal(F) + 0.05 * np.linspace(0, 1, F) ** 2).astype(np.float32)
    poly, fit = oracle.transpoly_fit(y, 6)
    A = oracle.transpoly_design(F, 6).astype(np.float64)
    ref = A @ np.linalg.lstsq(A, y.astype(np.float64), rcond=None)[0]
    assert np.max(np.abs(fit - ref)) < 5e-6


def test_paint_gain(oracle):
    cal = [1.5, -0.01, 2e-5, 0.3, 1e-3, -2e-6]
    T, P = 70.0, 1500.0
    want = cal[0] + cal[1] * T + cal[2] * T * T + (cal[3] + cal[4] * T + cal[5] * T * T) * P
    assert abs(oracle.paint_gain(cal, T, P) - want) < 1e-3 * abs(want)


def test_phase2_rows(oracle):
    rng = np.random.default_rng(11)
    n, F = 40, 300
    t = np.arange(F) / F
    I = (1000 + 50 * rng.standard_normal((n, F)) + 100 * t[None, :]).astype(np.float32)
    iref = I.mean(1).astype(np.float32)
    cov = np.ones(n, np.float32)
    cov[[3, 17]] = 0
    steady = (0.1 * rng.standard_normal(n)).astype(np.float32)
    temp = np.full(n, 65.0, np.float32)
    cal = [1.2, -0.004, 1e-5, 0.02, 1e-4, -1e-7]
    r = oracle.phase2(I, iref, cov, steady, temp, cal, qbar=250.0, ps=1800.0, degree=6, threads=2)
    assert np.isnan(r["sum"][3]) and np.isnan(r["gain"][17]) and np.isnan(r["pressure_t"][3]).all()
    # float64 restatement of the same formulas
    A = oracle.transpoly_design(F, 6).astype(np.float64)
    for i in (0, 5, 39):
        y = iref[i].astype(np.float64) / I[i].astype(np.float64)
        fit = A @ np.linalg.lstsq(A, y, rcond=None)[0]
        g = oracle.paint_gain(cal, 65.0, np.float32(250.0 * steady[i] + 1800.0))
        cp = (y - fit) * g * 144.0 / 250.0
        scale = np.abs(cp).max()
        assert np.max(np.abs(r["pressure_t"][i] - cp)) < 2e-4 * scale + 1e-5 * abs(g) * 144 / 250
        assert abs(r["sum"][i] - r["pressure_t"][i].astype(np.float64).sum()) < 1e-9 * F * scale
