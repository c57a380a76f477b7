"""GPU parity: per-frame operators and the frame loop vs the CPU oracle.
Bar: bit-exact for hot-pixel repair (integer), gather rows (float products are
IEEE-exact), transpose; accumulators relative 1e-12 (double, summation order)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_fix_hot_pixels(gpu_lib, oracle):
    import torch
    from upsp_processing_amd import engine
    rng = np.random.default_rng(11)
    F, H, W = 24, 37, 53          # ragged: not multiples of 8
    fr = rng.integers(0, 4000, size=(F, H, W)).astype(np.uint16)
    # 0..7 hot pixels per frame, some adjacent, some on corners/edges, some barely hot
    for f in range(F):
        for k in range(f % 8):
            fr[f, rng.integers(0, H), rng.integers(0, W)] = rng.choice([4064, 4095, 4070])
    fr[3, 0, 0] = 4095; fr[3, 0, 1] = 4095
    fr[5, H - 1, W - 1] = 4095; fr[5, H - 2, W - 1] = 4090
    fr[6, 10, 10] = 4095; fr[6, 10, 11] = 3900; fr[6, 9, 10] = 3900; fr[6, 11, 10] = 3900; fr[6, 10, 9] = 3900
    d = torch.as_tensor(fr).cuda()
    st = engine.fix_hot_pixels(d).cpu().numpy()
    out = d.cpu().numpy()
    for f in range(F):
        o, s = oracle.fix_hot_pixels(fr[f])
        assert s == st[f], f
        assert np.array_equal(o, out[f]), f
    assert (st == -1).any() and (st > 0).any()


def test_project_frame_and_transpose(gpu_lib, oracle):
    import torch
    from upsp_processing_amd import engine
    rng = np.random.default_rng(2)
    img = rng.integers(0, 4096, size=(64, 80)).astype(np.uint16)
    pix = rng.integers(-1, 64 * 80, size=5001).astype(np.int32)
    w = rng.random(5001).astype(np.float32)
    a = engine.project_frame(torch.as_tensor(img).cuda(), torch.as_tensor(pix).cuda(), torch.as_tensor(w).cuda())
    assert np.array_equal(a.cpu().numpy().view(np.int32), oracle.project_frame(img, pix, w).view(np.int32))
    imgf = rng.normal(size=(64, 80)).astype(np.float32)
    a = engine.project_frame(torch.as_tensor(imgf).cuda(), torch.as_tensor(pix).cuda(), None)
    assert np.array_equal(a.cpu().numpy().view(np.int32), oracle.project_frame(imgf, pix).view(np.int32))
    for shape in [(1, 1), (3, 129), (100, 77), (257, 64)]:
        m = rng.normal(size=shape).astype(np.float32)
        assert np.array_equal(engine.transpose(torch.as_tensor(m).cuda()).cpu().numpy(), m.T)


def run_loop_oracle(oracle, frames, pix, weight):
    """frame loop restated with the oracle pieces (psp_process.cpp:1771-1843)."""
    ncams, F = len(frames), frames[0].shape[0]
    n = pix.shape[1]
    sk = oracle.skipped_nodes(pix)
    s, ss = np.zeros(n), np.zeros(n)
    rows = np.zeros((F, n), np.float32)
    for f in range(F):
        sol = None
        for c in range(ncams):
            img, _ = oracle.fix_hot_pixels(frames[c][f])
            cs = oracle.project_frame(img, pix[c], weight[c])
            sol = cs if sol is None else (sol + cs).astype(np.float32)
        sol[sk] = np.nan
        oracle.accumulate(sol, s, ss)
        rows[f] = sol
    return rows, s, ss


@pytest.mark.parametrize("ncams", [1, 3])
def test_pipeline_rows_and_accumulators(gpu_lib, oracle, ncams):
    import torch
    from upsp_processing_amd import engine, synthetic as syn
    H, W, F, n = 96, 128, 41, 3000
    rng = np.random.default_rng(ncams)
    frames = [syn.synth_frames_numpy(F, H, W, seed=5 + c, hot=True) for c in range(ncams)]
    pix = rng.integers(-1, H * W, size=(ncams, n)).astype(np.int32)
    pix[:, ::17] = -1                                        # skipped in every camera
    weight = rng.random((ncams, n)).astype(np.float32) if ncams > 1 else np.ones((1, n), np.float32)
    rows_o, s_o, ss_o = run_loop_oracle(oracle, frames, pix, weight)

    pipe = engine.FramePipeline(ncams, W, H, n)
    for c in range(ncams):
        pipe.set_projection(c, pix[c], weight[c] if ncams > 1 else None)
    d_frames = [torch.as_tensor(f.copy()).cuda() for f in frames]
    rows_t = torch.zeros((n, F + 3), dtype=torch.float32, device="cuda")
    # two calls (ragged split) to exercise accumulation across calls + transposed output
    r1 = pipe.process([f[:17].contiguous() for f in d_frames], 0, rows_t=rows_t, col0=1)
    r2 = pipe.process([f[17:].contiguous() for f in d_frames], 17, rows_t=rows_t, col0=18)
    rows_g = torch.cat([r1, r2]).cpu().numpy()
    assert np.array_equal(rows_g.view(np.int32), rows_o.view(np.int32))
    assert np.array_equal(rows_t[:, 1:F + 1].cpu().numpy().view(np.int32), rows_o.T.view(np.int32))
    s_g, ss_g = [a.cpu().numpy() for a in pipe.accumulators()]
    ok = ~np.isnan(s_o)
    assert np.array_equal(np.isnan(s_g), np.isnan(s_o))
    assert np.allclose(s_g[ok], s_o[ok], rtol=1e-12) and np.allclose(ss_g[ok], ss_o[ok], rtol=1e-12)
    avg_g, rms_g = pipe.finalize(F)
    avg_o, rms_o = oracle.finals(s_o, ss_o, F)
    assert np.allclose(avg_g.cpu().numpy()[ok], avg_o[ok], rtol=1e-6)
    assert np.allclose(rms_g.cpu().numpy()[ok], rms_o[ok], rtol=1e-6)
    pipe.reset()
    assert pipe.accumulators()[0].abs().sum().item() == 0


def test_full_size_frame_loop_properties(gpu_lib):
    """BASELINE frame size (1024x1024, 0.5 M nodes): linearity + permutation properties.
    rows[f][n] must equal frame[f].flat[pix[n]] exactly, avg of a constant stack == the
    constant, transposed output == rows.T."""
    import torch
    from upsp_processing_amd import engine
    H = W = 1024
    n, F = 500766, 48
    g = torch.Generator(device="cuda"); g.manual_seed(3)
    pix = torch.randint(0, H * W, (n,), generator=g, device="cuda", dtype=torch.int32)
    pix[::5] = -1
    frames = torch.randint(0, 4000, (F, H, W), generator=g, device="cuda", dtype=torch.int32).to(torch.uint16)
    pipe = engine.FramePipeline(1, W, H, n)
    pipe.set_projection(0, pix)
    rows_t = torch.empty((n, F), dtype=torch.float32, device="cuda")
    rows = pipe.process(frames, 0, rows_t=rows_t)
    ref = frames.reshape(F, -1).to(torch.int32)[:, pix.clamp(min=0).long()].float()
    ref[:, pix < 0] = float("nan")
    assert torch.equal(rows.view(torch.int32), ref.view(torch.int32))
    assert torch.equal(rows_t.view(torch.int32), ref.t().contiguous().view(torch.int32))
    avg, rms = pipe.finalize(F)
    ok = pix >= 0
    assert torch.allclose(avg[ok], ref[:, ok].double().mean(0).float(), rtol=1e-6)
    assert torch.allclose(rms[ok], ref[:, ok].double().pow(2).mean(0).sqrt().float(), rtol=1e-6)


def test_hot_pixels_saturated_frames(gpu_lib, oracle):
    """Frames with very many pixels >= thresh (saturation) stay untouched (count > max_hot) and do
    not stall the scan; mixed with repairable frames in the same batch."""
    import torch
    from upsp_processing_amd import engine
    rng = np.random.default_rng(77)
    fr = rng.integers(100, 3000, (6, 256, 320)).astype(np.uint16)
    fr[1, 40:200, 10:300] = 4095                      # ~46k saturated pixels
    fr[3][rng.random((256, 320)) < 0.2] = 4090        # scattered
    fr[2, 17, 23] = 4095                              # repairable
    fr[4, 0, 0] = 4095
    fr[4, 255, 319] = 4080
    res = [oracle.fix_hot_pixels(fr[i]) for i in range(6)]
    want = np.stack([r[0] for r in res])
    wst = [int(r[1]) for r in res]
    d = torch.as_tensor(fr).cuda()
    st = engine.fix_hot_pixels(d).cpu().numpy()
    assert np.array_equal(d.cpu().numpy(), want)
    assert st.tolist() == wst and st[1] == -1 and st[3] == -1 and st[2] == 1


@pytest.mark.parametrize("fused", [1, 2])
@pytest.mark.parametrize("F,ld_extra", [(41, 0), (64, 4), (130, 3)])
def test_packed_and_u16_series(gpu_lib, oracle, F, ld_extra, fused):
    """Row map (packed series: only the rows of visible nodes are stored) and the u16 wire format
    of the time-series exchange, against the plain f32 node-major series -- bit for bit."""
    import torch
    from upsp_processing_amd import engine, synthetic as syn, _capi
    H, W, n = 96, 128, 3001
    rng = np.random.default_rng(F)
    frames = syn.synth_frames_numpy(F, H, W, seed=11, hot=True)
    frames[0].flat[:3] = (65535, 32768, 0)
    pix = rng.integers(0, H * W, size=n).astype(np.int32)
    pix[:3] = (0, 1, 2)
    pix[rng.random(n) < 0.4] = -1
    pix[:3] = (0, 1, 2)
    d_pix = torch.as_tensor(pix).cuda()
    pipe = engine.FramePipeline(1, W, H, n, hot_enable=0, fused_scan=fused)   # 1: fused pass, 2: scan + gather kernels
    pipe.set_projection(0, d_pix)
    d_frames = torch.as_tensor(frames).cuda()
    full = torch.empty((n, F), dtype=torch.float32, device="cuda")
    pipe.process(d_frames, 0, rows_t=full, want_rows=False)
    s0 = [a.clone() for a in pipe.accumulators()]
    # oracle: gather rows
    for f in (0, F - 1):
        sol = oracle.project_frame(frames[f], pix, None)
        sol[pix < 0] = np.nan
        assert np.array_equal(full[:, f].cpu().numpy().view(np.int32), sol.view(np.int32))
    vis = torch.nonzero(d_pix >= 0).reshape(-1)
    rowmap = torch.full((n,), -1, dtype=torch.int32, device="cuda")
    rowmap[vis] = torch.arange(vis.numel(), dtype=torch.int32, device="cuda")
    pipe.set_row_map(rowmap)
    ld = F + ld_extra
    for dtype in (torch.float32, torch.uint16):
        pipe.reset()
        if dtype == torch.float32:
            buf = torch.full((vis.numel(), ld), -7.0, dtype=dtype, device="cuda")
        else:
            buf = torch.full((vis.numel(), ld), 7, dtype=torch.int32, device="cuda").to(torch.uint16)
        half = 17
        pipe.process(d_frames[:half].contiguous(), 0, rows_t=buf[:, :F], want_rows=False)
        pipe.process(d_frames[half:].contiguous(), half, rows_t=buf[:, :F], col0=half, want_rows=False)
        got = buf.cpu().numpy()
        want = full.index_select(0, vis).cpu().numpy()
        assert np.array_equal(got[:, :F].astype(np.float32), want)
        assert (got[:, F:] == (7 if dtype == torch.uint16 else -7.0)).all()   # nothing past the frames
        s1 = pipe.accumulators()
        assert torch.equal(s0[0].view(torch.int64), s1[0].view(torch.int64))
        assert torch.equal(s0[1].view(torch.int64), s1[1].view(torch.int64))
        # receiving side: scatter the packed block back into a NaN-filled full series
        out = torch.full((n, F + 5), float("nan"), dtype=torch.float32, device="cuda")
        from upsp_processing_amd.distributed import _scatter_rows
        _scatter_rows(out, vis, 2, buf[:, :F].contiguous())
        assert torch.equal(out[:, 2:F + 2].contiguous().view(torch.int32), full.view(torch.int32))
        assert torch.isnan(out[:, :2]).all() and torch.isnan(out[:, F + 2:]).all()
    pipe.set_row_map(None)
    # u16 is refused wherever the stored values need not be 16-bit integers
    b16 = torch.zeros((n, F), dtype=torch.int32, device="cuda").to(torch.uint16)
    pipe.set_projection(0, d_pix, torch.ones(n))
    with pytest.raises(_capi.UpspError):
        pipe.process(d_frames, 0, rows_t=b16, want_rows=False)
    p2 = engine.FramePipeline(1, W, H, n, filter=1, filter_size=3)
    p2.set_projection(0, d_pix)
    with pytest.raises(_capi.UpspError):
        p2.process(d_frames, 0, rows_t=b16, want_rows=False)
    p3 = engine.FramePipeline(2, W, H, n)
    p3.set_projection(0, d_pix); p3.set_projection(1, d_pix)
    with pytest.raises(_capi.UpspError):
        p3.process([d_frames, d_frames.clone()], 0, rows_t=b16, want_rows=False)


@pytest.mark.parametrize("fused", [1, 2])
@pytest.mark.parametrize("F", [41, 150])
def test_hot_pixel_prescan_on_side_stream(gpu_lib, oracle, F, fused):
    """fix_hot_pixels queued ahead of the frame loop on a second stream (the schedule bench.py
    uses: the scan runs while the projection is built) + process(hot_fixed=True): rows,
    transposed rows, repaired frames and accumulators identical to the in-loop scan / the oracle."""
    import torch
    from upsp_processing_amd import engine, synthetic as syn
    H, W, n = 96, 128, 3000
    rng = np.random.default_rng(11)
    frames = [syn.synth_frames_numpy(F, H, W, seed=21, hot=True)]
    frames[0][3, 5, 7] = 4095
    frames[0][F - 1, H - 1, W - 1] = 4095
    frames[0][7][rng.random((H, W)) < 0.1] = 4090          # saturated frame: left alone
    pix = rng.integers(-1, H * W, size=(1, n)).astype(np.int32)
    pix[0, :200] = 5 * W + 7                                # many nodes on a repaired pixel
    weight = np.ones((1, n), np.float32)
    rows_o, s_o, ss_o = run_loop_oracle(oracle, frames, pix, weight)
    want_frames = np.stack([oracle.fix_hot_pixels(frames[0][f])[0] for f in range(F)])

    pipe = engine.FramePipeline(1, W, H, n, fused_scan=fused)
    d = torch.as_tensor(frames[0].copy()).cuda()
    side, main = torch.cuda.Stream(), torch.cuda.current_stream()
    side.wait_stream(main)
    with torch.cuda.stream(side):
        pipe.fix_hot_pixels(d)
    pipe.set_projection(0, pix[0])
    main.wait_stream(side)
    rows_t = torch.zeros((n, engine.series_ld(F)), dtype=torch.float32, device="cuda")
    rows = pipe.process(d, 0, rows_t=rows_t, hot_fixed=True)
    assert pipe.opts.hot_enable == 1
    assert np.array_equal(d.cpu().numpy(), want_frames)
    assert np.array_equal(rows.cpu().numpy().view(np.int32), rows_o.view(np.int32))
    assert np.array_equal(rows_t[:, :F].cpu().numpy().view(np.int32), rows_o.T.view(np.int32))
    s_g, ss_g = [a.cpu().numpy() for a in pipe.accumulators()]
    ok = ~np.isnan(s_o)
    assert np.array_equal(np.isnan(s_g), np.isnan(s_o))
    assert np.allclose(s_g[ok], s_o[ok], rtol=1e-12) and np.allclose(ss_g[ok], ss_o[ok], rtol=1e-12)
    # the in-loop scan on already repaired frames changes nothing (idempotent)
    pipe.reset()
    rows2 = pipe.process(d, 0)
    assert torch.equal(rows2.view(torch.int32), rows.view(torch.int32))


@pytest.mark.parametrize("fused", [1, 2])
@pytest.mark.parametrize("F", [70, 64])
def test_large_model_series_variants(gpu_lib, F, fused):
    """70 001 nodes, full u16 range: the node-major series written with and without the frame-major
    rows, in ragged calls, after a change of projection, packed as u16 through a row map and with an
    overlap source map -- all bit-identical to a torch gather; accumulators identical between the
    variants.  (Written for a pixel-sorted node order of the gather, which was measured and dropped:
    DESIGN.md section 4; kept as a consistency test of the gather's output paths.)"""
    import torch
    from upsp_processing_amd import engine
    H, W, n = 200, 300, 70001
    g = torch.Generator(device="cuda"); g.manual_seed(F)
    pix = torch.randint(0, H * W, (n,), generator=g, device="cuda", dtype=torch.int32)
    pix[torch.rand(n, generator=g, device="cuda") < 0.45] = -1
    pix[:5] = torch.tensor([H * W - 1, 0, -1, 0, H * W - 1], dtype=torch.int32, device="cuda")   # ties, extremes
    frames = torch.randint(0, 65536, (F, H, W), generator=g, device="cuda", dtype=torch.int32).to(torch.uint16)
    ref = frames.reshape(F, -1).to(torch.int32)[:, pix.clamp(min=0).long()].float()
    ref[:, pix < 0] = float("nan")
    pipe = engine.FramePipeline(1, W, H, n, hot_enable=0, fused_scan=fused)
    pipe.set_projection(0, pix)
    # with frame-major rows
    rt0 = torch.zeros((n, engine.series_ld(F)), dtype=torch.float32, device="cuda")
    rows = pipe.process(frames, 0, rows_t=rt0)
    assert torch.equal(rows.view(torch.int32), ref.view(torch.int32))
    s0 = [a.clone() for a in pipe.accumulators()]
    # node-major series only, two ragged calls
    pipe.reset()
    rt1 = torch.zeros_like(rt0)
    pipe.process(frames[:33].contiguous(), 0, rows_t=rt1, want_rows=False)
    pipe.process(frames[33:].contiguous(), 33, rows_t=rt1, col0=33, want_rows=False)
    assert torch.equal(rt1[:, :F].view(torch.int32), ref.t().contiguous().view(torch.int32))
    assert torch.equal(rt0.view(torch.int32), rt1.view(torch.int32))
    s1 = pipe.accumulators()
    assert torch.equal(s0[0].view(torch.int64), s1[0].view(torch.int64))
    assert torch.equal(s0[1].view(torch.int64), s1[1].view(torch.int64))
    # a new projection
    pix2 = pix.flip(0).contiguous()
    pipe.set_projection(0, pix2)
    pipe.reset()
    rt2 = torch.zeros_like(rt0)
    pipe.process(frames, 0, rows_t=rt2, want_rows=False)
    assert torch.equal(rt2[:, :F].view(torch.int32), ref.flip(1).t().contiguous().view(torch.int32))
    # packed u16 series through a row map + overlap source map
    vis = torch.nonzero(pix2 >= 0).reshape(-1)
    rowmap = torch.full((n,), -1, dtype=torch.int32, device="cuda")
    rowmap[vis] = torch.arange(vis.numel(), dtype=torch.int32, device="cuda")
    src = torch.arange(n, dtype=torch.int32, device="cuda")
    src[vis[:100]] = vis[100:200].to(torch.int32)          # the stored series of these nodes is another node's
    pipe.set_row_map(rowmap)
    pipe.set_overlap_source(src)
    pipe.reset()
    buf = torch.zeros((vis.numel(), engine.series_ld(F)), dtype=torch.int32, device="cuda").to(torch.uint16)
    pipe.process(frames, 0, rows_t=buf[:, :F], want_rows=False)
    want = rt2[:, :F].index_select(0, src.long()).index_select(0, vis)
    assert torch.equal(buf[:, :F].to(torch.int32).float(), want)
    s2 = pipe.accumulators()          # accumulators take the node's own value
    ok = pix2 >= 0
    assert torch.equal(s2[0][ok], rt2[:, :F].double().sum(1)[ok])


@pytest.mark.parametrize("F", [1, 5, 64, 200, 257, 300, 513, 1030])
def test_fused_pass_hot_pixels_vs_oracle(gpu_lib, oracle, F):
    """The fused scan + projection pass (fused_scan=1) against the oracle frame loop with hot pixels
    of every kind: repairable ones on pixels several nodes read, on pixels nobody reads, at the image
    corners and next to each other (the second repair sees the first), a frame with exactly max_hot,
    one with max_hot + 1 (left alone) and a saturated frame; node-major series, repaired frames and
    accumulators bit-identical, also for a user skip list and packed u16 rows."""
    import torch
    from upsp_processing_amd import engine, synthetic as syn
    H, W, n = 64, 130, 2500          # 8320 pixels: 65 tiles of 128
    rng = np.random.default_rng(100 + F)
    fr = syn.synth_frames_numpy(F, H, W, seed=33, hot=False)
    fr = np.minimum(fr, 3000).astype(np.uint16)
    pix = rng.integers(-1, H * W, size=(1, n)).astype(np.int32)
    pix[0, :40] = 7 * W + 9                       # many nodes on one pixel
    pix[0, 40:60] = 0                             # corner pixels
    pix[0, 60:80] = H * W - 1
    pix[0, 80] = 7 * W + 10
    f0 = 0
    fr[f0, 7, 9] = 4095; fr[f0, 7, 10] = 4090     # neighbours: the second repair reads the first
    fr[f0, 0, 0] = 4095; fr[f0, H - 1, W - 1] = 4070
    if F > 2:
        fr[2].flat[rng.choice(H * W, 5, replace=False)] = 4095         # exactly max_hot
        fr[1].flat[rng.choice(H * W, 6, replace=False)] = 4095         # one too many: untouched
    if F > 4:
        fr[4][rng.random((H, W)) < 0.3] = 4080                         # saturated
        fr[3, 20, 20] = 4064                                           # == thresh, small change vs neighbours
        fr[3, 19, 20] = fr[3, 21, 20] = fr[3, 20, 19] = fr[3, 20, 21] = 3900   # old - new = 164 <= 512: kept
    fr[F - 1, 30, 64] = 4095
    weight = np.ones((1, n), np.float32)
    rows_o, s_o, ss_o = run_loop_oracle(oracle, [fr], pix, weight)
    want_frames = np.stack([oracle.fix_hot_pixels(fr[f])[0] for f in range(F)])
    pipe = engine.FramePipeline(1, W, H, n, fused_scan=1)
    pipe.set_projection(0, pix[0])
    d = torch.as_tensor(fr.copy()).cuda()
    rt = torch.full((n, engine.series_ld(F)), -3.0, dtype=torch.float32, device="cuda")
    pipe.process(d, 0, rows_t=rt[:, :F], want_rows=False)
    assert np.array_equal(d.cpu().numpy(), want_frames)
    assert np.array_equal(rt[:, :F].cpu().numpy().view(np.int32), rows_o.T.view(np.int32))
    assert (rt[:, F:] == -3.0).all()
    s_g, ss_g = [a.cpu().numpy() for a in pipe.accumulators()]
    assert np.array_equal(s_g.view(np.int64), s_o.view(np.int64)) or (
        np.array_equal(np.isnan(s_g), np.isnan(s_o)) and np.array_equal(s_g[~np.isnan(s_o)], s_o[~np.isnan(s_o)]))
    ok = ~np.isnan(s_o)
    assert np.array_equal(ss_g[ok], ss_o[ok])
    # user skip list (some visible nodes skipped) + packed u16 rows
    sk = np.zeros(n, np.uint8); sk[pix[0] < 0] = 1; sk[5] = 1; sk[100] = 1
    pipe.set_skipped(torch.as_tensor(sk).cuda())
    rowmap = np.full(n, -1, np.int32); keep = np.nonzero(sk == 0)[0]; rowmap[keep] = np.arange(keep.size)
    pipe.set_row_map(torch.as_tensor(rowmap).cuda())
    pipe.reset()
    d2 = torch.as_tensor(fr.copy()).cuda()
    buf = torch.zeros((keep.size, engine.series_ld(F)), dtype=torch.int32, device="cuda").to(torch.uint16)
    pipe.process(d2, 0, rows_t=buf[:, :F], want_rows=False)
    assert np.array_equal(buf[:, :F].cpu().numpy().astype(np.float32), rows_o.T[keep])
    s2 = pipe.accumulators()[0].cpu().numpy()
    assert np.isnan(s2[5]) and np.isnan(s2[100]) and np.array_equal(s2[keep], s_o[keep])


@pytest.mark.parametrize("compact_mb", [0, 1])
def test_stuck_hot_pixels_whole_call(gpu_lib, oracle, compact_mb):
    """The case fix_hot_pixels exists for (cpp/utils/cv_extras.cpp:230-275): a camera with 5 stuck
    pixels, hot in every one of the 1000 frames of ONE process() call = 5 000 replaced pixels (round 1
    silently dropped everything past 4 096).  Two of the stuck pixels are read by many nodes, one by
    nobody, two sit next to each other (the second repair sees the first).  Node-major series,
    repaired frames and accumulators bit-identical to the oracle loop -- also with a compact-buffer
    budget that cuts the call into several frame groups, and packed u16 rows."""
    import torch
    from upsp_processing_amd import engine, synthetic as syn
    H, W, n, F = 48, 130, 2000, 1000
    rng = np.random.default_rng(4242)
    fr = syn.synth_frames_numpy(F, H, W, seed=77, hot=False)
    fr = np.minimum(fr, 3000).astype(np.uint16)
    stuck = [(7, 9), (7, 10), (30, 64), (0, 0), (H - 1, W - 1)]
    for (r, c) in stuck:
        fr[:, r, c] = 4095
    fr[:, 7, 10] = rng.integers(4064, 4096, F).astype(np.uint16)      # flickers between hot values
    fr[17, 20, 20] = 4095                                             # frame 17: six hot pixels -> left alone
    fr[500, 30, 64] = 2000                                            # frame 500: four hot pixels
    pix = rng.integers(-1, H * W, size=(1, n)).astype(np.int32)
    pix[0][pix[0] == 30 * W + 64] = -1
    pix[0, :300] = 7 * W + 9                       # many nodes on a stuck pixel
    pix[0, 300:450] = H * W - 1                    # ... and on another one
    pix[0, 450] = 7 * W + 10
    pix[0, 451] = 0
    weight = np.ones((1, n), np.float32)
    rows_o, s_o, ss_o = run_loop_oracle(oracle, [fr], pix, weight)
    want_frames = np.stack([oracle.fix_hot_pixels(fr[f])[0] for f in range(F)])
    assert (want_frames != fr).sum() > 4096        # more changes than round 1's list could hold
    pipe = engine.FramePipeline(1, W, H, n, fused_scan=1, compact_mb=compact_mb)
    pipe.set_projection(0, pix[0])
    d = torch.as_tensor(fr.copy()).cuda()
    rt = torch.full((n, engine.series_ld(F, whole_rows=True)), -3.0, dtype=torch.float32, device="cuda")
    pipe.process(d, 0, rows_t=rt[:, :F], want_rows=False)
    assert np.array_equal(d.cpu().numpy(), want_frames)
    assert np.array_equal(rt[:, :F].cpu().numpy().view(np.int32), rows_o.T.view(np.int32))
    assert (rt[:, F:] == -3.0).all()
    s_g, ss_g = [a.cpu().numpy() for a in pipe.accumulators()]
    ok = ~np.isnan(s_o)
    assert np.array_equal(np.isnan(s_g), np.isnan(s_o))
    assert np.array_equal(s_g[ok], s_o[ok]) and np.array_equal(ss_g[ok], ss_o[ok])
    # the same through the scan + gather schedule
    p2 = engine.FramePipeline(1, W, H, n, fused_scan=2)
    p2.set_projection(0, pix[0])
    d2 = torch.as_tensor(fr.copy()).cuda()
    rt2 = torch.full((n, engine.series_ld(F)), -3.0, dtype=torch.float32, device="cuda")
    p2.process(d2, 0, rows_t=rt2[:, :F], want_rows=False)
    assert torch.equal(rt2[:, :F].contiguous().view(torch.int32), rt[:, :F].contiguous().view(torch.int32))
    # packed u16 rows in two calls (second call starts at a column that is not a multiple of 64)
    keep = np.nonzero(pix[0] >= 0)[0]
    rowmap = np.full(n, -1, np.int32); rowmap[keep] = np.arange(keep.size)
    pipe.set_row_map(torch.as_tensor(rowmap).cuda())
    pipe.reset()
    d3 = torch.as_tensor(fr.copy()).cuda()
    buf = torch.zeros((keep.size, F + 8), dtype=torch.int32, device="cuda").to(torch.uint16)
    pipe.process(d3[:333].contiguous(), 0, rows_t=buf[:, :F], want_rows=False)
    pipe.process(d3[333:].contiguous(), 333, rows_t=buf[:, :F], col0=333, want_rows=False)
    assert np.array_equal(buf[:, :F].cpu().numpy().astype(np.float32), rows_o.T[keep])
    s3, ss3 = [a.cpu().numpy() for a in pipe.accumulators()]
    assert np.array_equal(s3[ok], s_o[ok]) and np.array_equal(ss3[ok], ss_o[ok])
    # u16 rows need a row map (NaN has no u16 encoding)
    pipe.set_row_map(None)
    from upsp_processing_amd import _capi
    with pytest.raises(_capi.UpspError):
        pipe.process(d3, 0, rows_t=torch.zeros((n, F), dtype=torch.int32, device="cuda").to(torch.uint16), want_rows=False)


@pytest.mark.parametrize("F", [41, 300])
def test_multi_camera_streamed_schedule(gpu_lib, oracle, F):
    """Three cameras with weights, node-major series only: the streamed schedule (fused_scan=1: one
    compact buffer per camera, pass B sums the cameras in order) against scan + gather (fused_scan=2)
    -- rows bit-identical, accumulators to 1e-12 (summation order) -- and against the oracle loop."""
    import torch
    from upsp_processing_amd import engine, synthetic as syn
    H, W, n, ncams = 96, 128, 3000, 3
    rng = np.random.default_rng(7 + F)
    frames = [syn.synth_frames_numpy(F, H, W, seed=50 + c, hot=True) for c in range(ncams)]
    pix = rng.integers(-1, H * W, size=(ncams, n)).astype(np.int32)
    pix[:, ::17] = -1                                        # skipped in every camera
    pix[1, ::5] = -1
    # stuck hot pixels in two cameras, both read by the same nodes (two replaced pixels on one (node, frame));
    # a third one next to the first (its repair sees the repaired neighbour)
    frames[0][:, 5, 7] = 4095
    frames[0][::2, 5, 8] = 4090
    frames[2][:, 9, 3] = 4095
    pix[0, 1:60] = 5 * W + 7
    pix[2, 1:60] = 9 * W + 3
    pix[0, 60:70] = 5 * W + 8
    weight = rng.random((ncams, n)).astype(np.float32)
    rows_o, s_o, ss_o = run_loop_oracle(oracle, frames, pix, weight)
    res = {}
    for mode in (1, 2):
        pipe = engine.FramePipeline(ncams, W, H, n, fused_scan=mode)
        for c in range(ncams):
            pipe.set_projection(c, pix[c], weight[c])
        d_frames = [torch.as_tensor(f.copy()).cuda() for f in frames]
        rt = torch.full((n, engine.series_ld(F)), -5.0, dtype=torch.float32, device="cuda")
        half = 17
        pipe.process([f[:half].contiguous() for f in d_frames], 0, rows_t=rt[:, :F], want_rows=False)
        pipe.process([f[half:].contiguous() for f in d_frames], half, rows_t=rt[:, :F], col0=half, want_rows=False)
        s, ss = [a.cpu().numpy() for a in pipe.accumulators()]
        res[mode] = (rt.cpu().numpy(), s, ss, [f.cpu().numpy() for f in d_frames])
    r1, r2 = res[1], res[2]
    assert np.array_equal(r1[0].view(np.int32), r2[0].view(np.int32))
    assert np.array_equal(r1[0][:, :F].view(np.int32), rows_o.T.view(np.int32))
    assert (r1[0][:, F:] == -5.0).all()
    for c in range(ncams):
        assert np.array_equal(r1[3][c], r2[3][c])            # frames repaired the same way
    ok = ~np.isnan(s_o)
    assert np.array_equal(np.isnan(r1[1]), np.isnan(s_o))
    assert np.allclose(r1[1][ok], s_o[ok], rtol=1e-12) and np.allclose(r1[2][ok], ss_o[ok], rtol=1e-12)
    assert np.allclose(r1[1][ok], r2[1][ok], rtol=1e-12) and np.allclose(r1[2][ok], r2[2][ok], rtol=1e-12)


def test_full_size_streamed_frame_loop_properties(gpu_lib):
    """BASELINE frame size (1024x1024, 0.5 M nodes) through the default streamed schedule, 300 frames
    (one pass A + one whole-row pass B): series == frame[pix] exactly, NaN rows for nodes without a
    pixel, accumulators == exact integer sums, identical to the scan + gather schedule."""
    import torch
    from upsp_processing_amd import engine
    H = W = 1024
    n, F = 500766, 300
    g = torch.Generator(device="cuda"); g.manual_seed(5)
    pix = torch.randint(0, H * W, (n,), generator=g, device="cuda", dtype=torch.int32)
    pix[::5] = -1
    pix[1:2000] = 77 * W + 99                       # many nodes on one pixel
    frames = torch.randint(0, 4000, (F, H, W), generator=g, device="cuda", dtype=torch.int32).to(torch.uint16)
    ld = engine.series_ld(F)
    out = {}
    for mode in (0, 2):
        pipe = engine.FramePipeline(1, W, H, n, fused_scan=mode)
        pipe.set_projection(0, pix)
        rt = torch.empty((n, ld), dtype=torch.float32, device="cuda")[:, :F]
        pipe.process(frames, 0, rows_t=rt, want_rows=False)
        out[mode] = (rt, [a.clone() for a in pipe.accumulators()])
    rt = out[0][0]
    vis = pix >= 0
    for f in (0, 63, 64, 255, 256, F - 1):
        ref = frames[f].reshape(-1).to(torch.int32)[pix.clamp(min=0).long()].float()
        assert torch.equal(rt[vis, f], ref[vis])
    assert torch.isnan(rt[~vis]).all()
    assert torch.equal(rt.contiguous().view(torch.int32), out[2][0].contiguous().view(torch.int32))
    s0, ss0 = out[0][1]
    assert torch.equal(s0[vis], rt[vis].double().sum(1)) and torch.equal(ss0[vis], (rt[vis] * rt[vis]).double().sum(1))
    assert torch.equal(s0.view(torch.int64), out[2][1][0].view(torch.int64))
    assert torch.equal(ss0.view(torch.int64), out[2][1][1].view(torch.int64))


@pytest.mark.parametrize("F,prepare", [(70, False), (300, False), (70, True), (300, True)])
def test_prescan_with_candidate_map(gpu_lib, oracle, F, prepare):
    """Pass A ahead of the projection (upsp_pipeline_set_active_hint + upsp_pipeline_prescan on a second
    stream), the projection set afterwards, process() = pass B + fix-up only: series, repaired frames and
    accumulators bit-identical to the oracle loop -- with hot pixels, with a projection that is a strict
    subset of the candidates, with nodes whose pixel is NOT among the candidates (served from the frames),
    and again after a projection change that reuses the candidate map.
    prepare: the projection arrives on a THIRD stream and upsp_pipeline_prepare_rows derives the nodes' series rows and the
    skipped flags there, beside pass A (the bench's default schedule); process() then launches pass B only."""
    import torch
    from upsp_processing_amd import engine, synthetic as syn
    H, W, n = 64, 130, 2500
    rng = np.random.default_rng(7 * F)
    fr = np.minimum(syn.synth_frames_numpy(F, H, W, seed=81, hot=False), 3000).astype(np.uint16)
    fr[:, 7, 9] = 4095                                   # stuck pixel, read by many nodes
    fr[3, 20, 21] = 4095
    cand = rng.integers(-1, H * W, size=n).astype(np.int32)
    cand[:40] = 7 * W + 9
    weight = np.ones((1, n), np.float32)
    pipe = engine.FramePipeline(1, W, H, n)
    if prepare:
        pipe.set_scan_split(True)            # (pass A in two launches, as the bench's default schedule runs it; 64 x 130 pixels = 65 tiles)
    side, main = torch.cuda.Stream(), torch.cuda.current_stream()
    for trial in range(2):
        pix = cand.copy()
        pix[rng.random(n) < 0.5] = -1                    # the rays saw only half of the candidates
        pix[:40] = 7 * W + 9
        extra = rng.choice(n, 25, replace=False)         # pixels outside the candidate set
        pix[extra] = rng.integers(0, H * W, 25)
        rows_o, s_o, ss_o = run_loop_oracle(oracle, [fr], pix[None], weight)
        want_frames = np.stack([oracle.fix_hot_pixels(fr[f])[0] for f in range(F)])
        d = torch.as_tensor(fr.copy()).cuda()
        side.wait_stream(main)
        with torch.cuda.stream(side):
            if trial == 0:
                pipe.set_active_hint(torch.as_tensor(cand).cuda())
            pipe.prescan(d)
        pipe.reset()
        if prepare:
            third = torch.cuda.Stream()
            third.wait_stream(main)
            with torch.cuda.stream(third):
                if trial == 0:
                    third.wait_stream(side)              # (the map the rows are looked up in)
                    pipe.set_projection(0, pix)
                else:
                    # the projection "built" straight into the pipeline's own buffer: taken over without a copy
                    tgt = pipe.projection_target(0)
                    tgt.copy_(torch.as_tensor(pix).cuda())
                    pipe.set_projection(0, tgt)
                    assert pipe.projection_target(0).data_ptr() != tgt.data_ptr()      # (the other buffer is handed out next)
                pipe.prepare_rows()
                tabs = pipe.row_tables()                 # what an exchange would be handed on this stream
                assert torch.equal(tabs["skipped"].bool().cpu(), torch.as_tensor(pix < 0))
                nk = tabs["node_k"].cpu().numpy()
                assert ((nk >= 0) | (nk == -2))[pix >= 0].all() and (nk[pix < 0] == -1).all()
                assert set(np.flatnonzero(nk == -2)) <= set(extra.tolist())          # only pixels outside the candidate set
            main.wait_stream(third)
        else:
            pipe.set_projection(0, pix)
        main.wait_stream(side)
        rt = torch.full((n, engine.series_ld(F, whole_rows=True)), -3.0, dtype=torch.float32, device="cuda")
        pipe.process(d, 0, rows_t=rt[:, :F], want_rows=False)
        assert np.array_equal(d.cpu().numpy(), want_frames)
        assert np.array_equal(rt[:, :F].cpu().numpy().view(np.int32), rows_o.T.view(np.int32))
        assert (rt[:, F:] == -3.0).all()
        s_g, ss_g = [a.cpu().numpy() for a in pipe.accumulators()]
        ok = ~np.isnan(s_o)
        assert np.array_equal(np.isnan(s_g), ~ok)
        assert np.array_equal(s_g[ok], s_o[ok]) and np.array_equal(ss_g[ok], ss_o[ok])
    # without a prescan the same pipeline still runs its own pass A (candidate map in place)
    pipe.reset()
    d = torch.as_tensor(fr.copy()).cuda()
    rt2 = torch.empty((n, F), dtype=torch.float32, device="cuda")
    pipe.process(d, 0, rows_t=rt2, want_rows=False)
    assert torch.equal(rt2.view(torch.int32), rt[:, :F].contiguous().view(torch.int32))
    pipe.set_active_hint(None)
    pipe.reset()
    d = torch.as_tensor(fr.copy()).cuda()
    rt3 = torch.empty((n, F), dtype=torch.float32, device="cuda")
    pipe.process(d, 0, rows_t=rt3, want_rows=False)
    assert torch.equal(rt3.view(torch.int32), rt2.view(torch.int32))


def test_stale_prescan_is_dropped(gpu_lib):
    """prescan(frames) -> set_projection(another projection, no active hint) -> process(frames): the compact
    series pass A wrote were laid out with the OLD active-pixel map and must not be consumed;
    the result equals a process call without any prescan.  Likewise after set_active_hint(None), and a
    prescan is dropped by an intervening process call of other frames."""
    import torch
    from upsp_processing_amd import engine, synthetic as syn
    H, W, n, F = 64, 130, 2500, 70
    rng = np.random.default_rng(5)
    fr = np.minimum(syn.synth_frames_numpy(F, H, W, seed=82, hot=False), 3000).astype(np.uint16)
    pix_a = rng.integers(-1, H * W, size=n).astype(np.int32)
    pix_b = rng.integers(-1, H * W, size=n).astype(np.int32)

    def plain(pix, frames):
        q = engine.FramePipeline(1, W, H, n)
        q.set_projection(0, pix)
        rt = torch.empty((n, F), dtype=torch.float32, device="cuda")
        q.process(torch.as_tensor(frames.copy()).cuda(), 0, rows_t=rt, want_rows=False)
        return rt.view(torch.int32).clone(), [a.clone() for a in q.accumulators()]

    want, acc = plain(pix_b, fr)
    d = torch.as_tensor(fr.copy()).cuda()
    for how in ("projection", "hint_cleared"):
        pipe = engine.FramePipeline(1, W, H, n)
        if how == "projection":
            pipe.set_projection(0, pix_a)
            pipe.prescan(d)
            pipe.set_projection(0, pix_b)
        else:
            pipe.set_active_hint(torch.as_tensor(pix_a).cuda())
            pipe.prescan(d)
            pipe.set_active_hint(None)
            pipe.set_projection(0, pix_b)
        rt = torch.empty((n, F), dtype=torch.float32, device="cuda")
        pipe.process(d, 0, rows_t=rt, want_rows=False)
        assert torch.equal(rt.view(torch.int32), want), how
        for g, w in zip(pipe.accumulators(), acc):
            assert torch.equal(g.view(torch.int64), w.view(torch.int64)), how
    # a prescan is consumed or dropped by the NEXT process call: other frames in between, then the first batch
    fr2 = np.minimum(syn.synth_frames_numpy(F, H, W, seed=83, hot=False), 3000).astype(np.uint16)
    pipe = engine.FramePipeline(1, W, H, n)
    pipe.set_projection(0, pix_b)
    pipe.prescan(d)
    d2 = torch.as_tensor(fr2.copy()).cuda()
    rt = torch.empty((n, F), dtype=torch.float32, device="cuda")
    pipe.process(d2, 0, rows_t=rt, want_rows=False)
    assert torch.equal(rt.view(torch.int32), plain(pix_b, fr2)[0])
    pipe.reset()
    pipe.process(d, 0, rows_t=rt, want_rows=False)
    assert torch.equal(rt.view(torch.int32), want)


def test_dropped_prescan_with_hot_pixels(gpu_lib, oracle):
    """A prescan repairs its frames (fix_hot_pixels, in place) and leaves the hot-pixel counters clean: when the next process call
    takes OTHER frames -- the prescan is dropped -- those are counted, repaired and projected as if no prescan had happened."""
    import torch
    from upsp_processing_amd import engine, synthetic as syn
    H, W, n, F = 64, 130, 2500, 70
    rng = np.random.default_rng(11)
    fr1 = syn.synth_frames_numpy(F, H, W, seed=90, hot=True)
    fr2 = syn.synth_frames_numpy(F, H, W, seed=91, hot=True)
    for fr in (fr1, fr2):
        fr[::3, 7, 9] = 4095                       # a hot pixel several nodes read, in every third frame
    pix = rng.integers(-1, H * W, size=n).astype(np.int32)
    pix[:30] = 7 * W + 9
    pipe = engine.FramePipeline(1, W, H, n)
    pipe.set_projection(0, pix)
    d1, d2 = torch.as_tensor(fr1.copy()).cuda(), torch.as_tensor(fr2.copy()).cuda()
    pipe.prescan(d1)
    assert np.array_equal(d1.cpu().numpy(), np.stack([oracle.fix_hot_pixels(f)[0] for f in fr1]))
    rt = torch.empty((n, F), dtype=torch.float32, device="cuda")
    pipe.process(d2, 0, rows_t=rt, want_rows=False)
    want_frames = np.stack([oracle.fix_hot_pixels(f)[0] for f in fr2])
    assert np.array_equal(d2.cpu().numpy(), want_frames)
    want = np.where(pix[:, None] >= 0, want_frames.reshape(F, -1)[:, np.maximum(pix, 0)].T.astype(np.float32), np.float32(np.nan))
    assert np.array_equal(rt.cpu().numpy().view(np.int32), want.view(np.int32))


def test_candidate_pixels_superset(gpu_lib):
    """upsp_projection_candidate_pixels: every pixel of the projection equals the node's candidate pixel."""
    import torch
    from upsp_processing_amd import _capi, engine, synthetic as syn
    v, t = syn.tunnel_model_quad(40, 14)
    s9, tn = syn.soup(v, t)
    nrm = syn.node_normals(v, t)
    for k1, az in ((0.0, 0), (-0.09, 40)):
        c = syn.pinhole_camera(512, 384, center=(0.1, 0.2, 20), half_extent=5.0, k1=k1, azimuth_deg=az)
        cam = _capi.make_camera(c["K"], c["dist"], c["R"], c["t"], 512, 384)
        bvh = engine.BVH(s9)
        pix = engine.build_projection(bvh, cam, v, nrm, tn, 70.0)["pix"]
        cand = engine.candidate_pixels(cam, v)
        seen = pix >= 0
        assert int(seen.sum()) > 100 and torch.equal(pix[seen], cand[seen])
        assert int((cand >= 0).sum()) > int(seen.sum())
        # ... restricted to the nodes that pass the oblique test (upsp_projection_candidate_pixels_oblique): still a superset,
        # exactly the nodes that cast a primary ray in the build, a subset of the plain candidates with the same pixels
        for angle in (70.0, 40.0):
            proj = engine.build_projection(bvh, cam, v, nrm, tn, angle, counts=False)
            pixa = proj["pix"]
            co = engine.candidate_pixels(cam, v, normals=nrm, oblique_angle_deg=angle)
            seen = pixa >= 0
            assert torch.equal(pixa[seen], co[seen])
            sub = co >= 0
            assert torch.equal(co[sub], cand[sub]) and int(sub.sum()) < int((cand >= 0).sum())
            assert int(sub.sum()) == engine.projection_counts(bvh)["primary_rays"]
        bvh.close()


def test_pixel_series_on_prescanned_candidates(gpu_lib, oracle):
    """upsp_pipeline_pixel_series after set_active_hint + prescan of the same frames (pass A beside the projection build, the
    N > 1 loop of bench.py): pass A is not repeated, the nodes get their rows in the candidate map's buffer, the hot pixels are
    repaired -- every node's series equals the repaired frame[pix], exactly like the series of a plain pixel_series call; a
    prescan of OTHER frames is not consumed (the plain path runs), and a candidate set that misses a pixel shows up as -2."""
    import torch
    from upsp_processing_amd import _capi, engine, synthetic as syn
    from upsp_processing_amd.engine import _DevArray
    v, t = syn.tunnel_model_quad(40, 14)
    s9, tn = syn.soup(v, t)
    nrm = syn.node_normals(v, t)
    W, H, F = 256, 192, 70
    c = syn.pinhole_camera(W, H, center=(0.1, 0.2, 20), half_extent=5.0)
    cam = _capi.make_camera(c["K"], c["dist"], c["R"], c["t"], W, H)
    bvh = engine.BVH(s9)
    pix = engine.build_projection(bvh, cam, v, nrm, tn, 70.0)["pix"]
    vis = torch.nonzero(pix >= 0).reshape(-1)
    frames0 = syn.synth_frames_numpy(F, H, W, seed=5, hot=False)
    p_hot = int(pix[vis[len(vis) // 2]].item())
    frames0[3].reshape(-1)[p_hot] = 4090                      # a hot pixel some node reads
    frames0[40].reshape(-1)[int(pix[vis[7]].item())] = 4085
    want = np.stack([oracle.fix_hot_pixels(f)[0] for f in frames0])

    def series_of(pipe, ps, d):
        comp = torch.as_tensor(_DevArray(ps["ptr"], ps["rows"] * ps["cpitch"], "<i2", pipe), device="cuda").view(ps["rows"], ps["cpitch"])
        nk = ps["node_k"].long()
        assert int((nk[vis] >= 0).all()) and int((nk[pix < 0] < 0).all())
        got = comp[nk[vis]][:, :F].cpu().numpy().view(np.uint16)                    # [visible nodes, F]
        assert np.array_equal(got.T, want.reshape(F, -1)[:, pix[vis].cpu().numpy()])
        assert np.array_equal(d.cpu().view(torch.int16).numpy().view(np.uint16), want)   # frames repaired in place

    for cand in (engine.candidate_pixels(cam, v), engine.candidate_pixels(cam, v, normals=nrm, oblique_angle_deg=70.0)):
        pipe = engine.FramePipeline(1, W, H, v.shape[0])
        d = torch.as_tensor(frames0.copy()).cuda()
        pipe.set_active_hint(cand)
        pipe.prescan(d)
        pipe.set_projection(0, pix)
        series_of(pipe, pipe.pixel_series(d), d)
        # a prescan of other frames is not consumed: the call falls back to the projection's own map and runs pass A
        d2 = torch.as_tensor(frames0.copy()).cuda()
        pipe.set_active_hint(cand)
        pipe.prescan(d)
        series_of(pipe, pipe.pixel_series(d2), d2)
        pipe.close()
    # candidates that miss pixels: the nodes on them get -2 (upsp_exchange_set_pixels refuses such a table)
    pipe = engine.FramePipeline(1, W, H, v.shape[0])
    d = torch.as_tensor(frames0.copy()).cuda()
    cand = engine.candidate_pixels(cam, v).clone()
    cand[vis[::2]] = -1
    pipe.set_active_hint(cand)
    pipe.prescan(d)
    pipe.set_projection(0, pix)
    nk = pipe.pixel_series(d)["node_k"]
    lost = ~torch.isin(pix[vis], cand[cand >= 0])
    assert int(lost.sum()) > 0 and bool((nk[vis][lost] == -2).all()) and bool((nk[vis][~lost] >= 0).all())
    pipe.close()
    bvh.close()


@pytest.mark.parametrize("ncams", [1, 3])
@pytest.mark.parametrize("F,chunks,extra", [(41, None, 0), (300, None, 0), (1000, None, 0), (1030, None, 0), (200, (70, 70, 60), 0),
                                            (200, (60, 70, 70), 0), (41, None, 128), (1000, None, 160), (200, (70, 70, 60), 128)])
def test_row_padding(gpu_lib, oracle, F, chunks, ncams, extra):
    """upsp_pipeline_set_row_padding: the row pass may write the columns between the last frame of a ROW and its pitch when they
    share a 128-byte line -- only a row's last line is ever padded.  Same frames through a pipeline with the padding declared and
    one without: the series [:, :F] and the accumulators bit-identical (and equal to the oracle loop); a chunk that ends inside the
    row (not in its last line) stores its own columns only, so does every call into a buffer whose pitch continues past the next
    boundary (extra > 0: a window of a wider matrix, live data to the right -- advisor finding, round 5)."""
    import torch
    from upsp_processing_amd import engine, synthetic as syn
    H, W, n = 64, 128, 2200
    rng = np.random.default_rng(F + ncams)
    frames = [np.minimum(syn.synth_frames_numpy(F, H, W, seed=70 + c, hot=False), 3000).astype(np.uint16) for c in range(ncams)]
    pix = rng.integers(-1, H * W, size=(ncams, n)).astype(np.int32)
    pix[:, ::13] = -1
    weight = rng.random((ncams, n)).astype(np.float32) if ncams > 1 else np.ones((1, n), np.float32)
    rows_o = run_loop_oracle(oracle, frames, pix, weight)[0] if F <= 300 else None
    ld = engine.series_ld(F, whole_rows=True) + extra         # (extra: the row goes on past the next 128-byte boundary)
    got = {}
    for pad in (False, True):
        pipe = engine.FramePipeline(ncams, W, H, n, fused_scan=1)
        for c in range(ncams):
            pipe.set_projection(c, pix[c], weight[c] if ncams > 1 else None)
        pipe.set_row_padding(pad)
        rt = torch.full((n, ld), -7.0, dtype=torch.float32, device="cuda")
        c0 = 0
        for nfr in (chunks or (F,)):
            d = [torch.as_tensor(f[c0:c0 + nfr].copy()).cuda() for f in frames]
            pipe.process(d, c0, rows_t=rt[:, :F], col0=c0, want_rows=False)
            c0 += nfr
        s, ss = [a.cpu().numpy() for a in pipe.accumulators()]
        got[pad] = (rt.cpu().numpy(), s, ss)
    plain, padded = got[False], got[True]
    assert np.array_equal(plain[0][:, :F].view(np.int32), padded[0][:, :F].view(np.int32))
    assert (plain[0][:, F:] == -7.0).all()
    stop = (F + 31) // 32 * 32
    assert (padded[0][:, stop:] == -7.0).all()
    tail = padded[0][:, F:stop]
    if stop > F and ncams == 1 and extra == 0:      # the padding was written: 0, or NaN in the row of a node no camera sees
        assert not (tail == -7.0).any() and (np.isnan(tail) | (tail == 0)).all()
    else:                            # (the several-camera row pass does not use the permission: measured slower there)
        assert (tail == -7.0).all()
    for a, b in zip(plain[1:], padded[1:]):
        assert np.array_equal(a.view(np.int64), b.view(np.int64))
    if rows_o is not None:
        assert np.array_equal(padded[0][:, :F].view(np.int32), rows_o.T.view(np.int32))


def test_row_padding_packed_u16_and_registration(gpu_lib):
    """The padding for the other whole-row writers: packed u16 series (128-byte line = 64 values) and registration as the last image
    stage (pass B over the warped active pixels)."""
    import torch
    from upsp_processing_amd import engine, synthetic as syn
    H, W, n, F = 64, 128, 1500, 100
    rng = np.random.default_rng(3)
    fr = np.minimum(syn.synth_frames_numpy(F, H, W, seed=9, hot=False), 3000).astype(np.uint16)
    pix = rng.integers(-1, H * W, size=n).astype(np.int32)
    keep = np.nonzero(pix >= 0)[0]
    rowmap = np.full(n, -1, np.int32); rowmap[keep] = np.arange(keep.size)
    out = {}
    for pad in (False, True):
        pipe = engine.FramePipeline(1, W, H, n, fused_scan=1)
        pipe.set_projection(0, pix)
        pipe.set_row_map(torch.as_tensor(rowmap).cuda())
        pipe.set_row_padding(pad)
        buf = torch.full((keep.size + 1, 128), 77, dtype=torch.int32, device="cuda").to(torch.uint16)
        pipe.process(torch.as_tensor(fr.copy()).cuda(), 0, rows_t=buf[:keep.size, :F], want_rows=False)
        out[pad] = buf.cpu().numpy()
    assert np.array_equal(out[False][:, :F], out[True][:, :F]) and (out[False][:, F:] == 77).all()
    assert (out[True][:keep.size, F:128] == 0).all() and (out[True][keep.size] == 77).all()
    assert np.array_equal(out[True][:keep.size, :F].T, fr.reshape(F, -1)[:, pix[keep]])
    out = {}
    for pad in (False, True):
        pipe = engine.FramePipeline(1, W, H, n, registration=1)
        pipe.set_projection(0, pix)
        pipe.set_reference(0, torch.as_tensor(fr[0].astype(np.float32)).cuda())
        pipe.set_row_padding(pad)
        rt = torch.full((n + 1, 128), -7.0, dtype=torch.float32, device="cuda")
        pipe.process(torch.as_tensor(fr.copy()).cuda(), 0, rows_t=rt[:n, :F], want_rows=False)
        out[pad] = rt.cpu().numpy()
    assert np.array_equal(out[False][:, :F].view(np.int32), out[True][:, :F].view(np.int32))
    assert (out[False][:, F:] == -7.0).all() and (out[True][n] == -7.0).all() and not (out[True][:n, F:128] == -7.0).any()


@pytest.mark.parametrize("sizes,pad", [([1000, 1000, 1000], 0), ([100, 40, 1000, 8], 24), ([36, 1024, 4, 0, 60], 0), ([1000], 24),
                                       ([37, 50, 3], 0), ([2048, 12], 20), ([8, 8, 8, 8], 0), ([64] * 40, 0), ([1000] * 8, 0)])
def test_rows_from_pixel_blocks(gpu_lib, sizes, pad):
    """upsp_rows_from_pixel_blocks: the owner's pass B over blocks as they arrive from the peers of an exchange ([pixel row][frames of
    the source] u16, one buffer per source).  Launches cut at 128-byte lines of the output rows read two blocks each; the rows and the
    accumulators must be what one long series buffer gives: row n = f32 of series[node_k[n]] (0 for a node without a pixel, NaN for a
    skipped one), sums exact.  Also blocks that are no multiple of 4 frames (one launch per block), empty blocks, blocks shorter than
    a line, blocks longer than one launch, more windows than one launch takes (16), padding columns at the end; with
    UPSP_ROWS_WINDOWS=0 / UPSP_ROWS_LINE_CUT=0 (tools/passb_probe.py) the same through one launch per window / per block."""
    import ctypes as C
    import torch
    from upsp_processing_amd import _capi
    rng = np.random.default_rng(sum(sizes) + pad)
    A, n = 700, 3000
    total = sum(sizes)
    series = rng.integers(0, 4096, size=(A, total)).astype(np.uint16)
    node_k = rng.integers(-1, A, size=n).astype(np.int32)
    skipped = (rng.random(n) < 0.1).astype(np.uint8)
    starts = np.concatenate([[0], np.cumsum(sizes)])
    pitch = [(sz + 3) // 4 * 4 for sz in sizes]              # (series rows on 8-byte boundaries)
    blocks = []
    for i, sz in enumerate(sizes):
        b = np.full((A, pitch[i]), 4444, np.uint16)
        b[:, :sz] = series[:, starts[i]:starts[i + 1]]
        blocks.append(torch.as_tensor(b).cuda() if sz else None)
    ld = (total + pad + 63) // 64 * 64 + 64
    rows = torch.full((n, ld), -9.0, dtype=torch.float32, device="cuda")
    s = torch.zeros(n, dtype=torch.float64, device="cuda"); ss = torch.zeros_like(s)
    ptrs = (C.c_void_p * len(sizes))(*[b.data_ptr() if b is not None else None for b in blocks])
    pitches = (C.c_uint32 * len(sizes))(*pitch)
    counts = (C.c_int64 * len(sizes))(*sizes)
    d_nk, d_sk = torch.as_tensor(node_k).cuda(), torch.as_tensor(skipped).cuda()
    for rep in range(2):       # twice: the accumulators add up
        _capi.check(_capi.lib().upsp_rows_from_pixel_blocks(ptrs, pitches, counts, len(sizes), C.c_void_p(d_nk.data_ptr()), C.c_void_p(d_sk.data_ptr()),
                                                            n, C.c_void_p(rows.data_ptr()), ld, total + pad, C.c_void_p(s.data_ptr()),
                                                            C.c_void_p(ss.data_ptr()), None))
    torch.cuda.synchronize()
    want = np.where(node_k[:, None] >= 0, series[np.maximum(node_k, 0)].astype(np.float32), np.float32(0))
    want[skipped != 0] = np.nan
    got = rows.cpu().numpy()
    assert np.array_equal(got[:, :total].view(np.int32), want.view(np.int32))
    assert (got[:, total + pad:] == -9.0).all()
    w64 = want.astype(np.float64)
    ws, wss = 2 * w64.sum(axis=1), 2 * (w64 * w64).sum(axis=1)
    gs, gss = s.cpu().numpy(), ss.cpu().numpy()
    ok = skipped == 0
    assert np.isnan(gs[~ok]).all() and np.isnan(gss[~ok]).all()
    assert np.array_equal(gs[ok], ws[ok]) and np.array_equal(gss[ok], wss[ok])


@pytest.mark.parametrize("F", [64, 200])
def test_pipeline_step_matches_plain_sequence(gpu_lib, oracle, F):
    """upsp_pipeline_step: one call per step of a frame loop that rebuilds its projection (model motion) -- build on the pipeline's own
    side stream, pass A on the candidate map beside it, hand-over, pass B, the previous step's finals, all events inside the
    library.  Four steps issued back to back without a host wait, the CAMERA moved every step (another projection, another
    candidate map), the frames of every step uploaded into the SAME device buffer by the frames hook (the point of the schedule
    where the previous step no longer reads them): series, repaired frames, accumulators and finals bit-identical to the plain
    one-stream sequence projection build -> set_projection -> reset -> process -> finalize on a second pipeline, and to the
    oracle loop for the first step."""
    import torch
    from upsp_processing_amd import _capi, engine, synthetic as syn
    v, t = syn.uv_sphere(40, 80)
    s9, tn = syn.soup(v, t)
    nrm = syn.node_normals(v, t)
    H = W = 128
    N = v.shape[0]
    bvh = engine.BVH(s9)
    d_v, d_n, d_tn = torch.as_tensor(v).cuda(), torch.as_tensor(nrm).cuda(), torch.as_tensor(tn).cuda()
    bvh.set_tri_nodes(d_tn, N)
    cams = []
    for s in range(4):
        c = syn.pinhole_camera(W, H, center=(0.3 * s, -0.2 * s, 20), half_extent=1.3, fill=0.8)
        cams.append(_capi.make_camera(c["K"], c["dist"], c["R"], c["t"], W, H))
    batches = [np.minimum(syn.synth_frames_numpy(F, H, W, seed=300 + s, hot=True), 4095).astype(np.uint16) for s in range(4)]
    staged = [torch.as_tensor(b).cuda() for b in batches]
    ld = engine.series_ld(F, whole_rows=True)
    # ---- the plain sequence, one stream ----
    ref = engine.FramePipeline(1, W, H, N)
    want = []
    for s in range(4):
        proj = engine.build_projection(bvh, cams[s], d_v, d_n, d_tn, 70.0, counts=False)
        ref.set_projection(0, proj["pix"])
        ref.reset()
        d = staged[s].clone()
        rt = torch.full((N, ld), -5.0, dtype=torch.float32, device="cuda")
        ref.process(d, first_frame=s * F, rows_t=rt[:, :F], want_rows=False)
        avg, rms = ref.finalize(F)
        want.append((rt[:, :F].clone(), d, avg, rms, proj["pix"].clone()))
    torch.cuda.synchronize()
    assert len({w[4].cpu().numpy().tobytes() for w in want}) > 1          # the projections do differ
    # ---- the same through upsp_pipeline_step ----
    pipe = engine.FramePipeline(1, W, H, N)
    pipe.set_row_padding(True)
    d = torch.empty_like(staged[0])
    rts = [torch.full((N, ld), -5.0, dtype=torch.float32, device="cuda") for _ in range(4)]
    fin = [(torch.empty(N, dtype=torch.float32, device="cuda"), torch.empty(N, dtype=torch.float32, device="cuda")) for _ in range(4)]
    after = []
    tails = []
    for s in range(4):
        def upload(st, s=s):
            if s:
                after.append(d.clone())              # (the previous step's repaired frames, before they are overwritten)
            d.copy_(staged[s])
        pipe.step(bvh, cams[s], d_v, d_n, d_tn, d, rows_t=rts[s][:, :F], first_frame=s * F, finals=fin[s], nframes_total=F,
                  frames_hook=upload, tail_hook=lambda st: tails.append(st.cuda_stream))
    pipe.step_finish()
    torch.cuda.synchronize()
    after.append(d.clone())
    # (the hooks run on the pipeline's side streams: two of them in turn, never the caller's)
    assert len(tails) == 4 and tails[0] == tails[2] and tails[1] == tails[3] and torch.cuda.current_stream().cuda_stream not in tails
    for s in range(4):
        rows, frames, avg, rms, _ = want[s]
        assert torch.equal(rts[s][:, :F].view(torch.int32), rows.view(torch.int32)), s
        assert torch.equal(after[s], frames), s
        assert torch.equal(fin[s][0].view(torch.int32), avg.view(torch.int32)) and torch.equal(fin[s][1].view(torch.int32), rms.view(torch.int32)), s
    # the first step against the oracle loop
    pix0 = want[0][4].cpu().numpy()
    rows_o, _, _ = run_loop_oracle(oracle, [batches[0]], pix0[None], np.ones((1, N), np.float32))
    assert np.array_equal(rts[0][:, :F].cpu().numpy().view(np.int32), rows_o.T.view(np.int32))
    # the pipeline is an ordinary pipeline afterwards
    pipe.set_active_hint(None)
    pipe.set_projection(0, want[3][4])
    pipe.reset()
    d2 = staged[3].clone()
    rt = torch.empty((N, F), dtype=torch.float32, device="cuda")
    pipe.process(d2, first_frame=3 * F, rows_t=rt, want_rows=False)
    assert torch.equal(rt.view(torch.int32), want[3][0].contiguous().view(torch.int32))
