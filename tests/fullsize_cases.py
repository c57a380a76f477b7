"""Per-rank full sizes of BASELINE configs[3] and configs[4] on ONE GPU (imported by tests/test_fullsize_gpu.py, and by
tools/full_size.py for the footprint / timing table of DESIGN.md section 3).

configs[3]: 100 000 frames on 8 GPUs -> 12 500 frames of 1024^2 per rank (26 GB resident), a [nodes_r, 100 000] series slice
            per rank; here one rank owns all 500 958 nodes for its 12 500 frames (W = 1, the N > 1 loop forced): the chunked
            pixel-series loop of psp.Phase1.frame_loop_pixel_wire (cpp/exec/psp_process.cpp:1519-1529, 1743-1851, 707-771).
configs[4]: 4 cameras x 50 000 frames on 8 GPUs -> 6 250 frame sets per rank on the 5 M-triangle model; here all 6 250 frame
            sets of 4 x 1024^2 (33.5 GB of frames resident) through the weighted multi-camera loop.
Everything is checked through size-independent properties: series == frame[pix] (camera-order weighted sum), NaN rows exactly
for the nodes no camera sees, exact integer sums, the hot-pixel repair of the frames that carry hot pixels against the oracle."""
import time

import numpy as np


def _hbm_used():
    import torch
    free, total = torch.cuda.mem_get_info()
    return total - free


def integer_frames(F, size, hot=None, seed=0, out=None):
    """u16 [F, size, size] on the device: a fixed integer ramp shifted per frame, values < 3500; hot: {frame: [(y, x), ...]} set to >= 4064."""
    import torch
    y = torch.arange(size, device="cuda", dtype=torch.int32)[:, None]
    x = torch.arange(size, device="cuda", dtype=torch.int32)[None, :]
    base = (x * 7 + y * 13 + seed * 101) % 3000
    fr = torch.empty((F, size, size), dtype=torch.uint16, device="cuda") if out is None else out
    for f0 in range(0, F, 250):
        n = min(250, F - f0)
        f = torch.arange(f0, f0 + n, device="cuda", dtype=torch.int32)[:, None, None]
        fr[f0:f0 + n] = ((base[None] + (f * 17) % 499) % 3500).to(torch.uint16)
    for f, lst in (hot or {}).items():
        for (yy, xx) in lst:
            fr[f, yy, xx] = 4095 - (yy % 16)
    return fr


def config3_rank_share(oracle, F=12500, size=1024, sample=3000, verbose=False):
    """Returns a dict of facts (and asserts the properties)."""
    import torch
    from upsp_processing_amd import distributed as D, psp, synthetic as syn
    t0 = time.time()
    used0 = _hbm_used()
    verts, tris = syn.tunnel_model_quad()
    s9, tn = syn.soup(verts, tris)
    cd = syn.pinhole_camera(size, size, center=(0, 0, 20), half_extent=6.0, fill=0.7)
    job = psp.Phase1(s9, tn, verts, syn.node_normals(verts, tris), [cd], (size, size), oblique_angle=70.0)
    N = verts.shape[0]
    pix = job.pix[0]
    vis = torch.nonzero(pix >= 0).reshape(-1)
    # hot pixels ON pixels some node reads, in the first, a middle and the last frame (one frame saturated: left alone)
    p_hot = [int(pix[vis[i]].item()) for i in (0, len(vis) // 2, len(vis) - 1, len(vis) // 3)]
    yx = [(p // size, p % size) for p in p_hot]
    hot = {0: [yx[0], yx[1]], F // 2 + 1: [yx[2]], F - 1: [yx[3], yx[0]], 7: [(i * 11, i * 7) for i in range(1, 9)]}
    frames = integer_frames(F, size, hot)
    originals = {f: frames[f].cpu().view(torch.int16).numpy().view(np.uint16).copy() for f in hot}
    used_frames = _hbm_used()
    shard = D.Shard(F, N, 0, 1)
    torch.cuda.synchronize()
    t1 = time.time()
    series = job.frame_loop_pixel_wire(shard, lambda c0, n: [frames[c0:c0 + n]], chunk=1024)
    torch.cuda.synchronize()
    t_loop = time.time() - t1
    used_peak = _hbm_used()
    assert tuple(series.shape) == (N, F)
    # the frames that carried hot pixels were repaired in place exactly like fix_hot_pixels (cv_extras.cpp:230-275)
    for f, orig in originals.items():
        want, _ = oracle.fix_hot_pixels(orig)
        got = frames[f].cpu().view(torch.int16).numpy().view(np.uint16)
        assert np.array_equal(got, want), f
        assert (want != orig).any() == (f != 7)          # frame 7 holds 8 > max_hot hot pixels: untouched
    sk = job.skipped.bool()
    assert bool(torch.isnan(series[sk]).all()) and not bool(torch.isnan(series[~sk]).any())
    # series == (repaired) frame[pix], all frames, for a sample of the nodes that see a pixel + every node on a repaired pixel
    g = torch.Generator(device="cuda"); g.manual_seed(5)
    pick = vis[torch.randperm(len(vis), generator=g, device="cuda")[:sample]]
    on_hot = torch.nonzero(torch.isin(pix, torch.tensor(p_hot, device="cuda", dtype=pix.dtype))).reshape(-1)
    pick = torch.unique(torch.cat([pick, on_hot]))
    flat = frames.view(torch.int16).reshape(F, -1)
    pp = pix[pick].long()
    bad = 0
    s_want = torch.zeros(len(pick), dtype=torch.float64, device="cuda")
    ss_want = torch.zeros(len(pick), dtype=torch.float64, device="cuda")
    for f0 in range(0, F, 500):
        want = flat[f0:f0 + 500][:, pp].to(torch.float32).T            # [pick, 500]
        bad += int((series[pick, f0:f0 + 500] != want).sum().item())
        s_want += want.double().sum(1)
        ss_want += (want * want).double().sum(1)
    assert bad == 0
    s, ss = job.pipe.accumulators()
    assert bool((s[pick] == s_want).all()) and bool((ss[pick] == ss_want).all())      # exact integer sums
    assert bool(torch.isnan(s[sk]).all())
    nodes_per_s = None
    facts = dict(frames=F, nodes=N, visible_nodes=int(len(vis)), exchange_chunks=D.chunk_count(shard.frame_count, 1024),
                 frames_GB=(used_frames - used0) / 1e9, peak_GB=(used_peak - used0) / 1e9,
                 series_slice_GB=series.shape[0] * series.stride(0) * 4 / 1e9, series_pitch_floats=int(series.stride(0)),
                 loop_seconds=t_loop, frames_per_s=F / t_loop, total_seconds=time.time() - t0)
    if verbose:
        print("configs[3] rank share:", facts, flush=True)
    job.close()
    return facts


def config4_rank_share(F=6250, size=1024, sample=1500, verbose=False):
    import torch
    from upsp_processing_amd import _capi, engine, synthetic as syn
    t0 = time.time()
    used0 = _hbm_used()
    v, t = syn.tunnel_model_quad(576, 205)                 # 4 990 104 triangles, 2 495 058 nodes
    s9, tn = syn.soup(v, t)
    nrm = syn.node_normals(v, t)
    N = v.shape[0]
    bvh = engine.BVH(s9)
    dn, dm, dt = [torch.as_tensor(x).cuda() for x in (v, nrm, tn)]
    bvh.set_tri_nodes(dt, N)
    cams, pix = [], []
    for az in (0, 90, 180, 270):
        cd = syn.pinhole_camera(size, size, center=(0, 0, 20), half_extent=6.0, azimuth_deg=az)
        cams.append(cd)
        cam = _capi.make_camera(cd["K"], cd["dist"], cd["R"], cd["t"], size, size)
        pix.append(engine.build_projection(bvh, cam, dn, dm, dt, 70.0, counts=False)["pix"])
    P = torch.stack(pix)
    centers = np.array([engine.camera_center(_capi.make_camera(c["K"], c["dist"], c["R"], c["t"], size, size)) for c in cams])
    w = engine.projection_weights(P, dn, dm, centers, "average_view")
    sk, ns = engine.skipped_nodes(P)
    frames = [integer_frames(F, size, {F // 3: [(100 + 10 * c, 200 + c)]}, seed=c) for c in range(4)]
    used_frames = _hbm_used()
    pipe = engine.FramePipeline(4, size, size, N)
    for c in range(4):
        pipe.set_projection(c, P[c], w[c])
    ld = engine.series_ld(F, whole_rows=True)
    rt = torch.empty((N, ld), dtype=torch.float32, device="cuda")[:, :F]
    torch.cuda.synchronize()
    t1 = time.time()
    for f0 in range(0, F, 1000):                            # calls of 1000 frame sets like a caller that reads its videos in pieces
        n = min(1000, F - f0)
        pipe.process([fr[f0:f0 + n] for fr in frames], f0, rows_t=rt, col0=f0, want_rows=False)
    torch.cuda.synchronize()
    t_loop = time.time() - t1
    used_peak = _hbm_used()
    skb = sk.bool()
    assert bool(torch.isnan(rt[skb]).all()) and not bool(torch.isnan(rt[~skb]).any())
    # series == sum over the cameras, in camera order, of w_c * f32(frame_c[pix_c])   (psp_process.cpp:1813-1819)
    g = torch.Generator(device="cuda"); g.manual_seed(6)
    seen = torch.nonzero(~skb).reshape(-1)
    two = torch.nonzero((P >= 0).sum(0) >= 2).reshape(-1)
    pick = torch.unique(torch.cat([seen[torch.randperm(len(seen), generator=g, device="cuda")[:sample]], two[:sample // 2]]))
    bad = 0
    s_want = torch.zeros(len(pick), dtype=torch.float64, device="cuda")
    for f0 in range(0, F, 500):
        sol = None
        for c in range(4):
            pc = P[c][pick]
            val = frames[c].view(torch.int16).reshape(F, -1)[f0:f0 + 500][:, pc.clamp(min=0).long()].to(torch.float32).T
            cs = torch.where((pc >= 0)[:, None], w[c][pick][:, None] * val, torch.zeros_like(val))
            sol = cs if sol is None else sol + cs
        bad += int((rt[pick, f0:f0 + 500] != sol).sum().item())
        s_want += sol.double().sum(1)
    assert bad == 0
    s, ss = pipe.accumulators()
    assert torch.allclose(s[pick], s_want, rtol=1e-12, atol=0)
    facts = dict(frame_sets=F, cameras=4, nodes=N, triangles=int(t.shape[0]), nodes_seen_by_two_or_more=int(len(two)),
                 frames_GB=(used_frames - used0) / 1e9, peak_GB=(used_peak - used0) / 1e9, series_GB=N * ld * 4 / 1e9,
                 loop_seconds=t_loop, frame_sets_per_s=F / t_loop, total_seconds=time.time() - t0)
    if verbose:
        print("configs[4] rank share:", facts, flush=True)
    pipe.close()
    bvh.close()
    return facts
