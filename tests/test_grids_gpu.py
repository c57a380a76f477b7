"""P3D zone overlaps in the frame loop (model.adjust_solution, psp_process.cpp:1827-1839): the
accumulators take a node's own value, the stored series the value of its source node.
Bit-exact vs a numpy restatement; plus the CLI on a two-zone PLOT3D grid."""
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("F,ncams", [(7, 1), (64, 2), (130, 1)])
def test_overlap_source_rows_and_sums(gpu_lib, F, ncams):
    import torch
    from upsp_processing_amd import engine
    rng = np.random.default_rng(F)
    H, W, N = 96, 128, 5000
    frames = [rng.integers(100, 4000, (F, H, W)).astype(np.uint16) for _ in range(ncams)]
    pix = [rng.integers(-1, H * W, N).astype(np.int32) for _ in range(ncams)]
    for p in pix:
        p[rng.random(N) < 0.3] = -1
    wts = [rng.random(N).astype(np.float32) for _ in range(ncams)] if ncams > 1 else [None]
    skipped = np.all(np.stack(pix) < 0, axis=0).astype(np.uint8)
    src = np.arange(N, dtype=np.int32)
    pick = rng.choice(np.arange(1, N), 400, replace=False)
    src[pick] = (rng.random(400) * pick).astype(np.int32)            # src[n] < n
    pipe = engine.FramePipeline(ncams, W, H, N, hot_enable=0)
    for c in range(ncams):
        pipe.set_projection(c, torch.as_tensor(pix[c]).cuda(), None if wts[c] is None else torch.as_tensor(wts[c]).cuda())
    pipe.set_skipped(torch.as_tensor(skipped).cuda())
    pipe.set_overlap_source(src)
    rows_t = torch.empty((N, engine.series_ld(F)), dtype=torch.float32, device="cuda")[:, :F]
    rows = pipe.process([torch.as_tensor(f).cuda() for f in frames], first_frame=0, rows_t=rows_t, want_rows=True)
    # numpy restatement: float sums in camera order, NaN, double accumulators BEFORE the copy
    sol = np.zeros((F, N), np.float32)
    for c in range(ncams):
        v = np.where(pix[c] >= 0, frames[c].reshape(F, -1)[:, np.maximum(pix[c], 0)].astype(np.float32), 0)
        w = np.float32(1) if wts[c] is None else wts[c][None, :]
        term = (np.float32(0) + w * v).astype(np.float32)
        term = np.where(pix[c] >= 0, term, np.float32(0))
        sol = term if c == 0 else (sol + term).astype(np.float32)
    sol = np.where(skipped[None, :] != 0, np.float32(np.nan), sol)
    want = sol[:, src]
    got = rows.cpu().numpy()
    assert np.array_equal(got.view(np.int32), want.view(np.int32))
    assert np.array_equal(rows_t.cpu().numpy().view(np.int32), want.T.view(np.int32))
    s, ss = pipe.accumulators()
    live = skipped == 0
    assert np.allclose(s.cpu().numpy()[live], sol.astype(np.float64).sum(0)[live], rtol=1e-13)
    assert np.allclose(ss.cpu().numpy()[live], (sol * sol).astype(np.float64).sum(0)[live], rtol=1e-13)
    # switching the map off restores the plain rows
    pipe.reset()
    pipe.set_overlap_source(None)
    rows2 = pipe.process([torch.as_tensor(f).cuda() for f in frames], first_frame=0, want_rows=True)
    assert np.array_equal(rows2.cpu().numpy().view(np.int32), sol.view(np.int32))
    pipe.close()


def test_cli_plot3d_grid(gpu_lib, tmp_path):
    """Two structured zones sharing an edge -> PLOT3D file -> psp_process: the nodes of the second
    zone on the shared edge carry the series of their partners in the first zone."""
    from upsp_processing_amd import grids, psp_process as cli, synthetic as syn
    from test_cli import write_case
    tmp = str(tmp_path)
    write_case(tmp, nframes=6)
    J, K = 24, 16
    u, v = np.meshgrid(np.linspace(-4, 0, J), np.linspace(-2, 2, K))            # zone 0: x in [-4, 0]
    u2, v2 = np.meshgrid(np.linspace(0, 4, J), np.linspace(-2, 2, K))           # zone 1: x in [0, 4]
    z = lambda a, b: 0.3 * np.cos(0.5 * a) + 0.1 * b
    g = dict(zones=[(J, K, 1), (J, K, 1)],
             x=np.concatenate([u.ravel(), u2.ravel()]).astype(np.float32),
             y=np.concatenate([v.ravel(), v2.ravel()]).astype(np.float32),
             z=np.concatenate([z(u, v).ravel(), z(u2, v2).ravel()]).astype(np.float32))
    grids.write_plot3d_grid(os.path.join(tmp, "model.x"), g)
    deck = open(os.path.join(tmp, "run.inp")).read().replace("model.tri", "model.x")
    open(os.path.join(tmp, "run.inp"), "w").write(deck)
    assert cli.main(["-input_file=%s/run.inp" % tmp, "-h5_out=x"]) == 0
    m = grids.P3DModel(g, 1e-3)
    src = m.overlap_source()
    moved = np.nonzero(src != np.arange(m.size()))[0]
    assert moved.size == K                                                       # the shared edge
    n, F = m.size(), 6
    series = np.fromfile(os.path.join(tmp, "out", "intensity_transpose"), "<f4").reshape(n, F)
    assert np.array_equal(series[moved].view(np.int32), series[src[moved]].view(np.int32))
    avg = np.fromfile(os.path.join(tmp, "out", "intensity_avg"), "<f4")
    assert np.array_equal(avg[moved].view(np.int32), avg[src[moved]].view(np.int32))
    assert np.isfinite(series).any()


def test_cli_plot3d_wind_on(gpu_lib, oracle, tmp_path):
    """PLOT3D model + steady-state function file (wind-on): gain per node from
    Pss = qbar * steady + ps (psp_process.cpp:2475-2476), float formula bit-exact."""
    import struct
    from upsp_processing_amd import grids, psp_process as cli
    from test_cli import write_case
    tmp = str(tmp_path)
    write_case(tmp, nframes=16)
    J, K = 20, 12
    u, v = np.meshgrid(np.linspace(-4, 4, J), np.linspace(-2, 2, K))
    g = dict(zones=[(J, K, 1)], x=u.ravel().astype(np.float32), y=v.ravel().astype(np.float32),
             z=(0.2 * np.cos(0.5 * u)).ravel().astype(np.float32))
    grids.write_plot3d_grid(os.path.join(tmp, "model.x"), g)
    n = J * K
    steady = (0.5 * np.sin(np.arange(n))).astype(np.float32)
    with open(os.path.join(tmp, "steady.f"), "wb") as f:          # no record separators
        f.write(struct.pack("<i", 1) + struct.pack("<iiii", J, K, 1, 1) + steady.tobytes())
    deck = open(os.path.join(tmp, "run.inp")).read().replace("model.tri", "model.x")
    deck = deck.replace("@all\n", "@all\n  sds = %s/run.wtd\n" % tmp)
    open(os.path.join(tmp, "run.inp"), "w").write(deck)
    open(os.path.join(tmp, "run.wtd"), "w").write("#  MACH TTF PS Q TCAVG\n0.85 95.0 1300.0 620.0 68.0\n")
    open(os.path.join(tmp, "paint.cal"), "w").write("a = 0.9\nb = -0.002\nd = 0.0008\n")
    assert cli.main(["-input_file=%s/run.inp" % tmp, "-h5_out=x", "-paint_cal=%s/paint.cal" % tmp,
                     "-steady_p3d=%s/steady.f" % tmp]) == 0
    out = os.path.join(tmp, "out")
    cov = np.fromfile(os.path.join(out, "coverage"), "<f4")
    gain = np.fromfile(os.path.join(out, "gain"), "<f4")
    live = cov != 0
    assert live.sum() > 20
    want = np.array([oracle.paint_gain([0.9, -0.002, 0, 0.0008, 0, 0], 68.0, np.float32(np.float32(620.0) * s + np.float32(1300.0)))
                     for s in steady], np.float32)
    assert np.array_equal(gain[live], want[live])
    assert np.array_equal(np.fromfile(os.path.join(out, "steady_state"), "<f4"), steady)
    with pytest.raises(cli.DeckError):
        cli.main(["-input_file=%s/run.inp" % tmp, "-h5_out=x", "-paint_cal=%s/paint.cal" % tmp,
                  "-steady_p3d=%s/paint.cal" % tmp])


def test_cli_tri_model_wind_on_interpolated(gpu_lib, oracle, tmp_path):
    """.tri model + structured steady grid: steady-state Cp interpolated onto the model nodes
    (upsp::interpolate, k = 10, p = 2) and written to steady_state."""
    import struct
    from upsp_processing_amd import grids, psp_process as cli
    from test_cli import write_case
    tmp = str(tmp_path)
    v, t, cams = write_case(tmp, nframes=8)
    J, K = 60, 30
    th, ph = np.meshgrid(np.linspace(0, 2 * np.pi, J), np.linspace(0.05, np.pi - 0.05, K))
    sg = dict(zones=[(J, K, 1)], x=(6 * np.cos(ph)).ravel().astype(np.float32),
              y=(np.sin(ph) * np.cos(th)).ravel().astype(np.float32), z=(np.sin(ph) * np.sin(th)).ravel().astype(np.float32))
    grids.write_plot3d_grid(os.path.join(tmp, "steady.x"), sg)
    cp = (0.5 * np.cos(ph) ** 2 - 0.2).ravel().astype(np.float32)
    with open(os.path.join(tmp, "steady.f"), "wb") as f:
        f.write(struct.pack("<i", 1) + struct.pack("<iiii", J, K, 1, 1) + cp.tobytes())
    deck = open(os.path.join(tmp, "run.inp")).read().replace("@all\n", "@all\n  sds = %s/run.wtd\n" % tmp)
    open(os.path.join(tmp, "run.inp"), "w").write(deck)
    open(os.path.join(tmp, "run.wtd"), "w").write("#  MACH TTF PS Q TCAVG\n0.85 95.0 1300.0 620.0 68.0\n")
    open(os.path.join(tmp, "paint.cal"), "w").write("a = 0.9\nb = -0.002\nd = 0.0008\n")
    args = ["-input_file=%s/run.inp" % tmp, "-h5_out=x", "-paint_cal=%s/paint.cal" % tmp, "-steady_p3d=%s/steady.f" % tmp]
    with pytest.raises(cli.DeckError):
        cli.main(args)                                              # .tri model: needs -steady_grid
    assert cli.main(args + ["-steady_grid=%s/steady.x" % tmp]) == 0
    nodes = np.stack([sg["x"], sg["y"], sg["z"]], axis=1)
    want, _ = oracle.interpolate_idw(nodes, cp, v, 10, 2.0)
    got = np.fromfile(os.path.join(tmp, "out", "steady_state"), "<f4")
    assert np.array_equal(got.view(np.int32), want.view(np.int32))
