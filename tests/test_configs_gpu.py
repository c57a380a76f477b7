"""GPU: the two multi-GPU shapes of BASELINE.json at sizes one GPU can hold.

configs[4]: 4 cameras, 5 M triangles (2.5 M nodes), frames x cameras -- projection of one camera against
the oracle at full mesh size, then the 4-camera weighted frame loop (AverageViews) checked through
size-independent properties (series == sum over the cameras, in camera order, of w_c * frame_c[pix_c];
NaN rows exactly for the nodes no camera sees; accumulators == sums of the stored series), and the two
schedules (streamed multi-camera / scan + gather) bit-identical.

configs[3]: frames sharded over ranks + the end-of-run exchange: two / three rank processes on this GPU, the
library's exchange bound to the tests' stand-in RCCL (tests/test_bench_gpu.py runs the bench that way); here the sharded result is compared with the
unsharded one on the same frames: series slices bit-identical, avg / rms identical (integer sums)."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_config4_shape_5m_triangles_4_cameras(gpu_lib, oracle):
    import torch
    from upsp_processing_amd import _capi, engine, synthetic as syn
    v, t = syn.tunnel_model_quad(576, 205)                 # 4 990 104 triangles, 2 495 058 nodes
    assert t.shape[0] > 4_900_000
    s9, tn = syn.soup(v, t)
    nrm = syn.node_normals(v, t)
    size, F = 1024, 32
    bvh = engine.BVH(s9)
    dn, dm, dt = [torch.as_tensor(x).cuda() for x in (v, nrm, tn)]
    bvh.set_tri_nodes(dt, v.shape[0])
    cams, pix = [], []
    for az in (0, 90, 180, 270):
        cd = syn.pinhole_camera(size, size, center=(0, 0, 20), half_extent=6.0, azimuth_deg=az)
        cams.append(cd)
        cam = _capi.make_camera(cd["K"], cd["dist"], cd["R"], cd["t"], size, size)
        pix.append(engine.build_projection(bvh, cam, dn, dm, dt, 70.0))
    # camera 1 (from the side: boosters occlude the body) against the oracle at the full mesh size
    obv = oracle.OracleBVH(s9)
    co = oracle.make_camera(cams[1]["K"], cams[1]["dist"], cams[1]["R"], cams[1]["t"], size, size)
    o = oracle.create_projection(obv, co, v, nrm, tn, engine.oblique_threshold(70.0), threads=min(os.cpu_count() or 1, 64))
    assert np.array_equal(pix[1]["pix"].cpu().numpy(), o["pix"])
    assert pix[1]["nrays"] == o["nrays"]
    assert np.array_equal(pix[1]["uv"].cpu().numpy().view(np.int32), o["uv"].view(np.int32))
    del obv
    P = torch.stack([p["pix"] for p in pix])
    centers = np.array([engine.camera_center(_capi.make_camera(c["K"], c["dist"], c["R"], c["t"], size, size)) for c in cams])
    w = engine.projection_weights(P, dn, dm, centers, "average_view")
    sk, ns = engine.skipped_nodes(P)
    assert 0 < ns < v.shape[0] and ((P >= 0).sum(0) >= 2).sum() > 1000          # overlap between cameras exists
    g = torch.Generator(device="cuda"); g.manual_seed(4)
    frames = [torch.randint(0, 4000, (F, size, size), generator=g, device="cuda", dtype=torch.int32).to(torch.uint16)
              for _ in range(4)]
    out = {}
    for mode in (1, 2):
        pipe = engine.FramePipeline(4, size, size, v.shape[0], hot_enable=0, fused_scan=mode)
        for c in range(4):
            pipe.set_projection(c, P[c], w[c])
        rt = torch.empty((v.shape[0], engine.series_ld(F)), dtype=torch.float32, device="cuda")[:, :F]
        pipe.process(frames, 0, rows_t=rt, want_rows=False)
        out[mode] = (rt, [a.clone() for a in pipe.accumulators()])
        pipe.close()
    rt = out[2][0]
    assert torch.equal(out[1][0].contiguous().view(torch.int32), rt.contiguous().view(torch.int32))
    for f in (0, F - 1):                                   # psp_process.cpp:1813-1819: sum in camera order
        sol = torch.zeros(v.shape[0], dtype=torch.float32, device="cuda")
        for c in range(4):
            pc = P[c]
            val = frames[c][f].reshape(-1).to(torch.int32)[pc.clamp(min=0).long()].float()
            term = torch.where(pc >= 0, 0.0 + w[c] * val, torch.zeros_like(val))
            sol = term if c == 0 else sol + term
        sol[sk] = float("nan")
        assert torch.equal(rt[:, f].contiguous().view(torch.int32), sol.view(torch.int32))
    assert torch.isnan(rt[sk]).all() and torch.isfinite(rt[~sk]).all()
    s, ss = out[2][1]
    ok = ~sk
    assert torch.allclose(s[ok], rt[ok].double().sum(1), rtol=1e-12)
    assert torch.allclose(ss[ok], (rt[ok] * rt[ok]).double().sum(1), rtol=1e-12)
    assert torch.allclose(out[1][1][0][ok], s[ok], rtol=1e-12)
    bvh.close()


def _rank_main(rank, world, port, tmp):
    import torch
    import torch.distributed as dist
    from upsp_processing_amd import distributed as D, engine
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    d = np.load(os.path.join(tmp, "case.npz"))
    pix, frames = torch.as_tensor(d["pix"]).cuda(), d["frames"]
    F, H, W = frames.shape
    N = pix.numel()
    shard = D.Shard(F, N, rank, world)
    f0, nf = shard.my_frames
    pipe = engine.FramePipeline(1, W, H, N)
    pipe.set_projection(0, pix)
    mine = torch.as_tensor(frames[f0:f0 + nf].copy()).cuda()
    # the chunked, packed, u16 exchange of bench.py: K chunks, all-to-all of chunk k in flight while k + 1 is processed
    K = 3
    exch = D.TimeSeriesExchange(shard, K)
    assert exch._x is not None and D.comm_ranks() == (rank, world)        # the library's exchange, not torch.distributed's
    exch.set_skipped(engine.skipped_nodes(pix, want_count=False)[0])
    pipe.set_row_map(exch.row_map())
    for k in range(K):
        c0, fc = exch.my_chunk(k)
        buf = torch.zeros((exch.packed_rows(), fc), dtype=torch.int32, device="cuda").to(torch.uint16)
        if fc:
            pipe.process(mine[c0:c0 + fc].contiguous(), first_frame=f0 + c0, rows_t=buf, want_rows=False)
        exch.submit(buf, packed=True)
    s, ss = pipe.accumulators()
    D.allreduce_sums(s, ss)
    series = exch.finish()
    avg, rms = pipe.finalize(F)
    n0, nn = shard.my_nodes
    # the same run with the ACTIVE PIXELS' series on the wire: pass A per chunk on the sender, pass B over all frames on the
    # owner of the nodes -- same series, same accumulators, bit for bit
    pipe2 = engine.FramePipeline(1, W, H, N)
    pipe2.set_projection(0, pix)
    mine2 = torch.as_tensor(frames[f0:f0 + nf].copy()).cuda()
    ex2 = D.TimeSeriesExchange(shard, K)
    tab = pipe2.pixel_series(None)
    ex2.set_pixels(tab["node_k"], engine.skipped_nodes(pix, want_count=False)[0])
    for k in range(K):
        c0, fc = ex2.my_chunk(k)
        ex2.submit_pixels(pipe2.pixel_series(mine2[c0:c0 + fc].contiguous()) if fc else tab)
    s2, ss2 = pipe2.accumulators()
    series2 = ex2.finish_pixels(s2, ss2)
    D.allreduce_sums(s2, ss2)
    assert torch.equal(series2.view(torch.int32), series.view(torch.int32))
    ok = ~torch.isnan(s)
    assert torch.equal(torch.isnan(s2), ~ok) and torch.equal(s2[ok], s[ok]) and torch.equal(ss2[ok], ss[ok])
    assert torch.equal(mine2, mine)                                   # frames repaired in place either way
    rows_px, rows_in = ex2.pixel_rows()
    assert rows_px <= 2 * int(tab["nactive"].item()) and rows_px < exch.packed_rows()
    np.savez(os.path.join(tmp, "rank%d.npz" % rank), series=series.cpu().numpy(), avg=avg.cpu().numpy(),
             rms=rms.cpu().numpy(), n0=n0, nn=nn)
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_config3_shape_sharded_frames_two_ranks(gpu_lib, oracle, rccl_shim, tmp_path, monkeypatch, world):
    import socket
    import torch
    import torch.multiprocessing as mp
    from upsp_processing_amd import engine, synthetic as syn
    H, W, N, F = 128, 160, 9001, 333
    rng = np.random.default_rng(9)
    frames = syn.synth_frames_numpy(F, H, W, seed=3, hot=True)
    frames[:, 5, 7] = 4095                                  # a stuck pixel some nodes read
    pix = rng.integers(-1, H * W, N).astype(np.int32)
    pix[:50] = 5 * W + 7
    np.savez(str(tmp_path / "case.npz"), pix=pix, frames=frames)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.get_context("spawn")
    # both ranks on cuda:0: the library's exchange binds the tests' stand-in RCCL (the children inherit the environment)
    monkeypatch.setenv("UPSP_RCCL_LIBRARY", rccl_shim)
    mp.spawn(_rank_main, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    # unsharded reference on the same frames
    pipe = engine.FramePipeline(1, W, H, N)
    pipe.set_projection(0, torch.as_tensor(pix).cuda())
    rt = torch.empty((N, F), dtype=torch.float32, device="cuda")
    pipe.process(torch.as_tensor(frames.copy()).cuda(), 0, rows_t=rt, want_rows=False)
    avg, rms = pipe.finalize(F)
    full = rt.cpu().numpy()
    for r in range(world):
        d = np.load(str(tmp_path / ("rank%d.npz" % r)))
        n0, nn = int(d["n0"]), int(d["nn"])
        assert np.array_equal(d["series"].view(np.int32), full[n0:n0 + nn].view(np.int32))
        assert np.array_equal(d["avg"].view(np.int32), avg.cpu().numpy().view(np.int32))
        assert np.array_equal(d["rms"].view(np.int32), rms.cpu().numpy().view(np.int32))
    # ... and the ORACLE loop on the same frames (fix_hot_pixels -> project_frame -> NaN rows -> double accumulators,
    # psp_process.cpp:1771-1843): what the two ranks deliver after the exchange is the reference's series, bit for bit
    sk = oracle.skipped_nodes(pix[None])
    want = np.empty((F, N), np.float32)
    s_o, ss_o = np.zeros(N), np.zeros(N)
    for f in range(F):
        img, _ = oracle.fix_hot_pixels(frames[f])
        sol = oracle.project_frame(img, pix, None)
        sol[sk] = np.nan
        oracle.accumulate(sol, s_o, ss_o)
        want[f] = sol
    got = np.concatenate([np.load(str(tmp_path / ("rank%d.npz" % r)))["series"] for r in range(world)])
    assert np.array_equal(got.view(np.int32), want.T.view(np.int32))
    a_o, r_o = (s_o / F).astype(np.float32), np.sqrt(ss_o / F).astype(np.float32)
    ok = ~sk
    assert np.isnan(avg.cpu().numpy()[sk]).all()
    assert np.array_equal(avg.cpu().numpy()[ok].view(np.int32), a_o[ok].view(np.int32))
    assert np.allclose(rms.cpu().numpy()[ok], r_o[ok], rtol=1e-6, atol=0)
