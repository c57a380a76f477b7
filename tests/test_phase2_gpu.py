"""Phase 2 (node-major delta-Cp) on the GPU vs the oracle and vs float64 least squares.

Tolerances (floating point, stated): the reference fits with a float QR (Eigen, un-vendored);
the GPU projects onto an orthogonal basis in double.  Both are compared with the float64
least-squares fit; the GPU must be within 3e-7 relative of it (the reference's own KAT allows
1e-4 absolute on values ~10, cpp/test/test_filtering.cpp:22), and within 1e-5*|y| of the oracle
and of the float64 solve on the reference's float-rounded design matrix.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _series(n, F, seed):
    rng = np.random.default_rng(seed)
    t = np.arange(F) / F
    I = (1500 + 40 * rng.standard_normal((n, F)) + 120 * t[None, :] - 60 * t[None, :] ** 3)
    return I.astype(np.float32)


def test_transpolyfitter_kat_gpu(gpu_lib):
    import torch
    from upsp_processing_amd import engine
    from test_phase2_oracle import _kat_inputs
    degree, n_frames, n_pts, y = _kat_inputs()
    fitter = engine.TransPolyFitter(n_frames, degree, n_pts)
    c_pts = 3                                              # blocks of 3 points like the KAT
    for p in range(0, n_pts, c_pts):
        chunk = y[p:p + c_pts]
        fit = fitter.eval_fit(torch.as_tensor(chunk), chunk.shape[0], p).cpu().numpy()
        assert np.max(np.abs(fit - chunk)) < 1e-4          # test_filtering.cpp:22,78
    # coefficients reproduce 2.5/(c+1) + p/(c+1) (loosely: the monomial problem is ill-conditioned)
    poly = fitter.poly.cpu().numpy()
    x = np.arange(n_frames) / n_frames
    for p in (0, 7, 12):
        assert np.max(np.abs(np.polyval(poly[p][::-1].astype(np.float64), x) - y[p])) < 1e-3


@pytest.mark.parametrize("F", [5, 64, 300, 1000, 1024, 1025, 3000])
def test_fit_vs_lstsq(gpu_lib, oracle, F):
    import torch
    from upsp_processing_amd import engine
    n = 37
    y = (_series(n, F, F) / 1500).astype(np.float32)
    fit = engine.TransPolyFitter(F, 6, n).eval_fit(torch.as_tensor(y)).cpu().numpy()
    A = oracle.transpoly_design(F, 6).astype(np.float64)       # float-rounded powers (reference)
    t = 2.0 * np.arange(F) / F - 1.0
    Ax = np.stack([t ** c for c in range(7)], axis=1)         # same polynomial space, exact abscissae
    for i in range(0, n, 6):
        yi = y[i].astype(np.float64)
        exact = Ax @ np.linalg.lstsq(Ax, yi, rcond=None)[0]
        assert np.max(np.abs(fit[i] - exact)) < 3e-7 * np.abs(exact).max()
        # rounding the powers to float (what the reference fits with) moves the fit by ~1e-6
        ref = A @ np.linalg.lstsq(A, yi, rcond=None)[0]
        assert np.max(np.abs(fit[i] - ref)) < 1e-5 * np.abs(ref).max()
        # the float QR of the reference carries its own rounding noise (grows with F: float dot
        # products over F samples); it must bracket the truth at the reference's KAT tolerance
        # (1e-4 on values ~10 -> 1e-5 relative ... allow 1e-4 at F = 3000) and the GPU fit must be
        # at least as close to the exact fit as the oracle is
        _, ofit = oracle.transpoly_fit(y[i], 6)
        err_orc = np.max(np.abs(ofit - exact))
        assert err_orc < 1e-4 * np.abs(exact).max()
        assert np.max(np.abs(fit[i] - exact)) <= err_orc + 1e-7


@pytest.mark.parametrize("F,ld", [(300, 300), (1000, 1000), (1500, 2048)])
def test_phase2_vs_oracle(gpu_lib, oracle, F, ld):
    import torch
    from upsp_processing_amd import engine
    n = 203
    I = _series(n, F, 3)
    rng = np.random.default_rng(9)
    iref = I.astype(np.float64).mean(1).astype(np.float32)
    cov = np.ones(n, np.float32)
    cov[[0, 50, 202]] = 0
    steady = (0.2 * rng.standard_normal(n)).astype(np.float32)
    temp = (60 + 10 * rng.random(n)).astype(np.float32)
    cal = [1.2, -0.004, 1e-5, 0.02, 1e-4, -1e-7]
    qbar, ps = 250.0, 1800.0
    want = oracle.phase2(I, iref, cov, steady, temp, cal, qbar, ps, 6)
    buf = torch.zeros((n, ld), dtype=torch.float32, device="cuda")
    buf[:, :F] = torch.as_tensor(I)
    got = engine.phase2_pressure(buf[:, :F], iref, cov, cal, qbar, ps, steady=steady, model_temp=temp)
    P = got["pressure_t"].cpu().numpy()
    live = cov != 0
    assert np.isnan(P[~live]).all()
    assert np.isnan(got["avg"].cpu().numpy()[~live]).all() and np.isnan(got["gain"].cpu().numpy()[~live]).all()
    # gain: float formula, bit-exact
    assert np.array_equal(got["gain"].cpu().numpy()[live], want["gain"][live].astype(np.float32))
    # delta-Cp: fit difference (<= 2e-5 of y ~ 1) times gain * 144 / qbar
    scale = np.abs(want["gain"][live]).max() * 144.0 / qbar
    assert np.max(np.abs(P[live] - want["pressure_t"][live])) < 2e-5 * scale
    # reductions agree with the device rows themselves to double rounding
    s = P[live].astype(np.float64).sum(1)
    ss = (P[live] * P[live]).astype(np.float64).sum(1)
    assert np.allclose(got["sum"].cpu().numpy()[live], s, rtol=0, atol=1e-9 * F * scale)
    assert np.allclose(got["sumsq"].cpu().numpy()[live], ss, rtol=1e-7, atol=0)   # exact vs float-rounded squares
    assert np.allclose(got["rms"].cpu().numpy()[live], np.sqrt(ss / F), rtol=1e-6)
    # in place
    got2 = engine.phase2_pressure(buf[:, :F], iref, cov, cal, qbar, ps, steady=steady, model_temp=temp,
                                  out=buf[:, :F])
    assert np.array_equal(buf[:, :F].cpu().numpy()[live], P[live])
    assert torch.equal(got2["sum"][torch.as_tensor(live)], got["sum"][torch.as_tensor(live)])


def test_phase2_wind_off_scalar_temp(gpu_lib, oracle):
    import torch
    from upsp_processing_amd import engine
    n, F = 64, 500
    I = _series(n, F, 21)
    iref = I.mean(1).astype(np.float32)
    cov = np.ones(n, np.float32)
    cal = [0.9, 0.0, 0.0, 0.01, 0.0, 0.0]
    want = oracle.phase2(I, iref, cov, np.zeros(n, np.float32), np.full(n, 72.5, np.float32), cal, 300.0, 2000.0, 6)
    got = engine.phase2_pressure(torch.as_tensor(I).cuda(), iref, cov, cal, 300.0, 2000.0, model_temp=72.5)
    scale = np.abs(want["gain"]).max() * 144.0 / 300.0
    assert np.max(np.abs(got["pressure_t"].cpu().numpy() - want["pressure_t"])) < 2e-5 * scale


def test_phase2_errors(gpu_lib):
    import torch
    from upsp_processing_amd import engine
    I = torch.ones((4, 10), dtype=torch.float32, device="cuda")
    with pytest.raises(Exception):
        engine.phase2_pressure(I, np.ones(4), np.ones(4), [1, 0, 0, 0, 0, 0], 1.0, 1.0, degree=9)
    with pytest.raises(ValueError):
        engine.phase2_pressure(I, np.ones(3), np.ones(4), [1, 0, 0, 0, 0, 0], 1.0, 1.0)


def test_phase2_full_size_properties(gpu_lib):
    """BASELINE size (500 958 nodes x 1000 frames): size-independent properties.
    * a series that IS a polynomial of degree <= 6 in f/F is removed completely (delta-Cp ~ 0);
    * delta-Cp is linear in the gain and invariant under scaling I and Iref together;
    * rms^2 == mean(cp^2) of the stored rows."""
    import torch
    from upsp_processing_amd import engine
    n, F = 500958, 1000
    g = torch.Generator(device="cuda").manual_seed(7)
    t = torch.arange(F, device="cuda", dtype=torch.float64) / F
    c = torch.rand((n, 4), device="cuda", generator=g, dtype=torch.float64)
    poly = 1.0 + 0.05 * c[:, :1] * t[None] - 0.04 * c[:, 1:2] * t[None] ** 3 + 0.02 * c[:, 2:3] * t[None] ** 6
    I = (1500.0 / poly).to(torch.float32)                     # Iref / I = poly (up to float rounding)
    iref = torch.full((n,), 1500.0, device="cuda")
    cov = torch.ones(n, device="cuda")
    cal = [1.0, 0, 0, 0, 0, 0]
    r = engine.phase2_pressure(I, iref, cov, cal, 144.0, 1000.0)          # gain 1, 144/q = 1
    assert float(r["pressure_t"].abs().max()) < 5e-6                       # float rounding of y only
    # linear in the gain, invariant under a common scale of I and Iref
    noise = 1.0 + 0.01 * torch.randn((n, F), device="cuda", generator=g)
    I2 = (I * noise).contiguous()
    a = engine.phase2_pressure(I2, iref, cov, cal, 144.0, 1000.0)
    b = engine.phase2_pressure(I2, iref, cov, [2.0, 0, 0, 0, 0, 0], 144.0, 1000.0)
    assert torch.equal(b["pressure_t"], 2.0 * a["pressure_t"])             # power of two: exact
    s = engine.phase2_pressure((I2 * 2.0).contiguous(), iref * 2.0, cov, cal, 144.0, 1000.0)
    assert torch.equal(s["pressure_t"], a["pressure_t"])
    ms = (a["pressure_t"].double() ** 2).mean(1)
    assert torch.allclose(a["rms"].double() ** 2, ms, rtol=1e-5, atol=1e-12)
    assert float(a["rms"].mean()) > 1e-3
