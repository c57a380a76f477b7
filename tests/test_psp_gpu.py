"""GPU: phase-1 driver (upsp_processing_amd/psp.py) vs the same phase restated with the
oracle pieces in reference order (psp_process.cpp:1597-1979): two cameras, AverageViews
weights, skipped nodes, sol1 / intensity_ratio_0, coverage, flat output files."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_phase1_two_cameras(gpu_lib, oracle, tmp_path):
    import torch
    from upsp_processing_amd import engine, psp, synthetic as syn
    W, H, F = 256, 192, 23
    v, t = syn.tunnel_model_quad(24, 10)
    s9, tn = syn.soup(v, t)
    nrm = syn.node_normals(v, t)
    cams = [syn.pinhole_camera(W, H, center=(0.3, 0.1, 20), half_extent=6.5, azimuth_deg=az)
            for az in (0, 55)]
    frames = [syn.synth_frames_numpy(F, H, W, seed=21 + c, hot=True) for c in range(2)]

    job = psp.Phase1(s9, tn, v, nrm, cams, (W, H), oblique_angle=70.0, overlap="average_view")
    finals, series = psp.run_phase1(job, [f.copy() for f in frames], out_dir=str(tmp_path), chunk=10)

    # ---- oracle, reference order --------------------------------------------------
    obv = oracle.OracleBVH(s9)
    thr = engine.oblique_threshold(70.0)
    ocams = [oracle.make_camera(c["K"], c["dist"], c["R"], c["t"], W, H) for c in cams]
    pix = np.stack([oracle.create_projection(obv, oc, v, nrm, tn, thr)["pix"] for oc in ocams])
    centers = np.array([oracle.cam_center(oc) for oc in ocams])
    wgt = oracle.adjust_weights(pix, np.ones_like(pix, dtype=np.float32), v, nrm, centers, 1)
    sk = oracle.skipped_nodes(pix)
    assert np.array_equal(job.pix.cpu().numpy(), pix)
    assert ((pix >= 0).sum(0) == 2).sum() > 20
    n = v.shape[0]

    def solve(imgs):
        sol = None
        for c in range(2):
            cs = oracle.project_frame(imgs[c], pix[c], wgt[c])
            sol = cs if sol is None else (sol + cs).astype(np.float32)
        sol[sk] = np.nan
        return sol

    sol1 = solve([oracle.fix_hot_pixels(frames[c][0])[0] for c in range(2)])
    s, ss = np.zeros(n), np.zeros(n)
    rows = np.zeros((F, n), np.float32)
    for f in range(F):
        rows[f] = solve([oracle.fix_hot_pixels(frames[c][f])[0] for c in range(2)])
        oracle.accumulate(rows[f], s, ss)
    avg, rms = oracle.finals(s, ss, F)
    cov = solve([np.ones((H, W), np.float32)] * 2)
    cov[sk] = (oracle.project_frame(np.ones((H, W), np.float32), pix[0], wgt[0]) +
               oracle.project_frame(np.ones((H, W), np.float32), pix[1], wgt[1]))[sk]   # no NaN in coverage

    ok = ~sk
    # weights differ by <= 1 ulp (f64 acos -> f32), so rows agree to 2 ulp of the sum
    assert np.allclose(series.cpu().numpy()[ok], rows.T[ok], rtol=3e-7, atol=0)
    assert np.isnan(series.cpu().numpy()[sk]).all()
    assert np.allclose(finals["avg"].cpu().numpy()[ok], avg[ok], rtol=1e-6)
    assert np.allclose(finals["rms"].cpu().numpy()[ok], rms[ok], rtol=1e-6)
    assert np.allclose(job.sol1.cpu().numpy()[ok], sol1[ok], rtol=3e-7)
    with np.errstate(all="ignore"):
        ratio = (avg / sol1 - 1.0).astype(np.float32)
    assert np.allclose(finals["ratio_0"].cpu().numpy()[ok], ratio[ok], rtol=0, atol=2e-6)
    assert np.allclose(finals["coverage"].cpu().numpy(), cov, rtol=3e-7)
    # flat files: raw little-endian f32, no header
    tr = np.fromfile(os.path.join(tmp_path, "intensity_transpose"), dtype="<f4").reshape(n, F)
    assert np.array_equal(tr.view(np.int32), series.cpu().numpy().view(np.int32))
    assert np.fromfile(os.path.join(tmp_path, "intensity_avg"), dtype="<f4").size == n
    assert np.fromfile(os.path.join(tmp_path, "cam02-uv"), dtype="<f4").size == 2 * n
    assert np.fromfile(os.path.join(tmp_path, "vv-int-avg.dat"), dtype="<f4").size == min(n, 1000)
    job.close()


def test_phase1_registration_filter_two_cameras(gpu_lib, oracle):
    """Two cameras with pixel registration + Gaussian filter through the phase-1 driver vs the
    oracle pieces in reference order (template = RAW first frame for the loop, repaired first
    frame for sol1; frame 0 is not registered in the loop, psp_process.cpp:1662-1679, 1777)."""
    import torch
    from upsp_processing_amd import engine, psp, synthetic as syn
    W, H, F = 224, 160, 7
    v, t = syn.tunnel_model_quad(16, 6)
    s9, tn = syn.soup(v, t)
    nrm = syn.node_normals(v, t)
    cams = [syn.pinhole_camera(W, H, center=(0.3, 0.1, 20), half_extent=6.5, azimuth_deg=az) for az in (0, 55)]
    frames = [syn.synth_frames_numpy(F, H, W, seed=31 + c, noise=2.0, hot=True) for c in range(2)]
    job = psp.Phase1(s9, tn, v, nrm, cams, (W, H), overlap="average_view", registration=True,
                     filter="gaussian", filter_size=3)
    finals, series = psp.run_phase1(job, [f.copy() for f in frames], chunk=4)

    obv = oracle.OracleBVH(s9)
    thr = engine.oblique_threshold(70.0)
    ocams = [oracle.make_camera(c["K"], c["dist"], c["R"], c["t"], W, H) for c in cams]
    pix = np.stack([oracle.create_projection(obv, oc, v, nrm, tn, thr)["pix"] for oc in ocams])
    centers = np.array([oracle.cam_center(oc) for oc in ocams])
    wgt = oracle.adjust_weights(pix, np.ones_like(pix, dtype=np.float32), v, nrm, centers, 1)
    sk = oracle.skipped_nodes(pix)
    n = v.shape[0]
    rows = np.zeros((F, n), np.float32)
    for f in range(F):
        sol = None
        for c in range(2):
            img, _ = oracle.fix_hot_pixels(frames[c][f])
            if f > 0:
                img, M, it = oracle.register_pixel(frames[c][0].astype(np.float32), img)
                assert it > 0
            img = oracle.blur(img.astype(np.float32), 3)
            cs = oracle.project_frame(img, pix[c], wgt[c])
            sol = cs if sol is None else (sol + cs).astype(np.float32)
        rows[f] = sol
    ok = ~sk
    got = series.cpu().numpy()
    assert np.isnan(got[sk]).all()
    # frame 0: no registration -> only the 1-ulp camera weights differ
    assert np.allclose(got[ok, 0], rows[0, ok], rtol=3e-7)
    d = np.abs(got[ok, 1:] - rows[1:, ok].T)
    assert d.max() <= 12.0 and d.mean() <= 0.5        # same bar as tests/test_imageops_gpu.py
    job.close()
