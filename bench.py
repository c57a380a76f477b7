#!/usr/bin/env python3
"""bench.py -- BASELINE.json metric on MI355X.

One STEP = one pass of the hot path over one batch of synthetic input
(BASELINE.json configs[1]: 1 000 frames, 1024x1024, ~1 M-triangle wind-tunnel model,
ray cast + projection only):

    projection build  (create_projection_mat: ~0.5 M node rays + jitter retries)
  + frame loop        (hot-pixel repair -> nearest-pixel projection -> double
                       accumulators -> node-major time series) over every frame
  + finals            (avg / rms)
  + for N > 1: the end-of-run exchanges (all_reduce of the accumulators and the
    all_to_all time-series exchange over RCCL/xGMI)

with mesh, BVH, camera and all frames already resident in HBM.  Frames shard over
ranks (weak scaling: every GPU processes --frames frames).

Prints ONE JSON line on rank 0.  `value` = frames/s over the whole job;
`mrays_per_s` = node rays/s of the projection-build kernel alone.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--frames", type=int, default=1000, help="frames per GPU per step")
    ap.add_argument("--size", type=int, default=1024, help="frame is size x size")
    ap.add_argument("--small", action="store_true", help="reduced mesh/frames (plumbing check)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-reraycast", action="store_true", help="skip the re-raycast stress figure (profiling runs)")
    ap.add_argument("--registration", action="store_true",
                    help="configs[2] shape: per-frame ECC registration before the projection")
    ap.add_argument("--force-chunked", action="store_true",
                    help="run the chunked / pipelined-exchange frame loop of the N>1 path on one GPU")
    ap.add_argument("--f32-wire", action="store_true",
                    help="N>1 / --force-chunked: exchange the series as f32 instead of u16")
    ap.add_argument("--overlap", action="store_true",
                    help="hot-pixel scan of all frames on a side stream, concurrent with the projection "
                         "build (measured slower on MI355X: the gathers then miss the Infinity Cache)")
    ap.add_argument("--two-kernel", action="store_true",
                    help="frame loop as scan kernel + gather kernel instead of the (default) streamed two-pass "
                         "schedule (scan + compact pixel series, then one pass over the nodes)")
    ap.add_argument("--model", default="quad", choices=["quad", "uv"],
                    help="quad: cube-sphere tunnel model (valence <= 6); uv: UV-sphere model with "
                         "1000-valent polar fans (worst case for per-ray traversal length)")
    return ap.parse_args()


def cpu_baseline(verts, tris, cam_dict, size, nframes_step, pix_full, sample_frames=256):
    """Oracle (CPU restatement, kind 'port') on a bounded sample of the same workload."""
    from concurrent.futures import ThreadPoolExecutor
    from oracle import oracle as orc
    from upsp_processing_amd import synthetic as syn, engine
    cores = os.cpu_count() or 1
    s9, tn = syn.soup(verts, tris)
    nrm = syn.node_normals(verts, tris)
    t0 = time.perf_counter()
    obv = orc.OracleBVH(s9)
    t_build = time.perf_counter() - t0
    cam = orc.make_camera(cam_dict["K"], cam_dict["dist"], cam_dict["R"], cam_dict["t"], size, size)
    # projection build on a node sample (every k-th node), all cores (OpenMP)
    k = 1
    dn = np.zeros(verts.shape[0], np.uint8)
    dn[::k] = 1
    t0 = time.perf_counter()
    r = orc.create_projection(obv, cam, verts, nrm, tn, engine.oblique_threshold(70.0), datanode=dn,
                              threads=cores)
    t_proj = time.perf_counter() - t0
    mrays = r["nrays"] / t_proj / 1e6
    # the frame sample gathers through the complete projection (the index array the
    # GPU step produced -- input data for the timed CPU loop, parity-checked in tests/)
    pix = np.ascontiguousarray(pix_full, dtype=np.int32)
    frames = syn.synth_frames_numpy(sample_frames, size, size, seed=99)
    sk = orc.skipped_nodes(pix)

    def one(f):
        img, _ = orc.fix_hot_pixels(frames[f])
        sol = orc.project_frame(img, pix, None)
        sol[sk] = np.nan
        return sol

    s, ss = np.zeros(pix.size), np.zeros(pix.size)
    t0 = time.perf_counter()
    with ThreadPoolExecutor(cores) as ex:
        for sol in ex.map(one, range(sample_frames)):
            orc.accumulate(sol, s, ss)
    t_frames = time.perf_counter() - t0
    per_frame = t_frames / sample_frames
    t_proj_full = t_proj * k
    fps = nframes_step / (t_proj_full + per_frame * nframes_step)
    return {"value": fps, "unit": "frames/s", "cores": cores, "kind": "port",
            "sample": "oracle/: projection build on every %d-th node (%d rays, %.2f s, OpenMP) "
                      "+ %d frames of the frame loop (%.3f s); extrapolated to the %d-frame step"
                      % (k, r["nrays"], t_proj, sample_frames, t_frames, nframes_step),
            "mrays_per_s": mrays, "frame_loop_frames_per_s": 1.0 / per_frame,
            "bvh_build_s": t_build}


def main():
    a = parse()
    import torch
    import torch.distributed as dist
    from upsp_processing_amd import _capi, engine, synthetic as syn, distributed as D

    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (no CPU fallback)")
    # rehearsal of the N > 1 code path on a one-GPU box: UPSP_BENCH_BACKEND=gloo with every rank on
    # cuda:0 (UPSP_BENCH_ONE_GPU=1).  The driver's runs use RCCL, one rank per GPU.
    backend = os.environ.get("UPSP_BENCH_BACKEND", "nccl")
    if os.environ.get("UPSP_BENCH_ONE_GPU"):
        local = 0
    torch.cuda.set_device(local)
    if world > 1:
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world,
                                    device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    _capi.lib()

    size = a.size
    F = a.frames
    if a.small:
        verts, tris = syn.tunnel_model_quad(64, 24)
        F = min(F, 64)
    elif a.model == "uv":
        verts, tris = syn.tunnel_model()          # 1 001 520 triangles, 500 766 nodes, polar fans
    else:
        verts, tris = syn.tunnel_model_quad()     # 1 001 904 triangles, 500 958 nodes, valence <= 6
    s9, tn = syn.soup(verts, tris)
    nrm = syn.node_normals(verts, tris)
    N = verts.shape[0]
    cd = syn.pinhole_camera(size, size, center=(0, 0, 20), half_extent=6.0)
    cam = _capi.make_camera(cd["K"], cd["dist"], cd["R"], cd["t"], size, size)

    bvh = engine.BVH(s9)
    d_nodes = torch.as_tensor(verts).cuda()
    d_nrm = torch.as_tensor(nrm).cuda()
    d_tn = torch.as_tensor(tn).cuda()
    bvh.set_tri_nodes(d_tn, N)      # createBVH(model, triNodes): once per model, like the BVH itself
    # this rank's frames (global frame index = rank*F + f), resident in HBM
    frames = torch.empty((F, size, size), dtype=torch.uint16, device="cuda")
    chunk = 50
    for f0 in range(0, F, chunk):
        syn.synth_frames_torch(min(chunk, F - f0), size, size, first=rank * F + f0, out=frames[f0:f0 + chunk])
    shard = D.Shard(F * world, N, rank, world)
    pipe = engine.FramePipeline(1, size, size, N, registration=int(a.registration),
                                fused_scan=2 if a.two_kernel else 0)
    if a.registration:
        pipe.set_reference(0, frames[0].to(torch.float32))   # raw first frame as ECC template
    # node-major time series [N, F] with the padded row pitch engine.series_ld() recommends
    ld = int(os.environ.get("UPSP_BENCH_LD", "0")) or engine.series_ld(F, whole_rows=True)
    rows_t = (torch.empty((N, ld), dtype=torch.float32, device="cuda")[:, :F]
              if not (world > 1 or a.force_chunked) else None)
    torch.cuda.synchronize()

    ev = lambda: torch.cuda.Event(enable_timing=True)
    t_ray, t_frames, t_xchg = [], [], []
    nrays_last = [0]
    primary_rays_last, retry_nodes_last = [0], [0]

    # N > 1: the frame loop runs in K chunks and the all-to-all of chunk k is issued
    # asynchronously while chunk k+1 is being processed (distributed.TimeSeriesExchange)
    chunked = world > 1 or a.force_chunked          # --force-chunked: exercise the N>1 loop on one GPU
    K = 4 if chunked else 1
    exch = D.TimeSeriesExchange(shard, K) if chunked else None
    chunk_bufs = ([torch.empty((N, exch.my_chunk(k)[1]), dtype=torch.float32, device="cuda")
                   for k in range(K)] if chunked else None)
    # one camera, no weights, no filter: the series values are exact 16-bit integers, so the
    # travelling rows are produced and sent as u16 (half the bytes) and widened by the receiver
    chunk_bufs16 = ([torch.empty((N, exch.my_chunk(k)[1]), dtype=torch.uint16, device="cuda")
                     for k in range(K)] if chunked else None)

    ev_log = []
    mode = {"packed": True, "u16": not a.f32_wire}
    # fix_hot_pixels does not depend on the projection: with --overlap its streaming scan of every
    # frame runs on a side stream while the (latency-bound) projection build occupies the main
    # stream and the gathers wait for both.  Same results (tests/test_frames_gpu.py).  Measured on
    # MI355X it LOSES (step 1.68 -> 1.86 ms): the gathers then read frames from HBM instead of the
    # Infinity Cache (39 -> 59 us per launch) and the traversal kernels slow down by a third under
    # the scan's traffic -- so it is off by default.
    overlap = a.overlap and not a.registration
    side = torch.cuda.Stream() if overlap else None

    def step(record):
        e = [ev() for _ in range(4)]
        e[0].record()
        main = torch.cuda.current_stream()
        if overlap:
            side.wait_stream(main)
            with torch.cuda.stream(side):
                pipe.fix_hot_pixels(frames)
        proj = engine.build_projection(bvh, cam, d_nodes, d_nrm, d_tn, 70.0, counts=False)   # no host sync
        e[1].record()
        pipe.reset()
        pipe.set_projection(0, proj["pix"])
        if overlap:
            main.wait_stream(side)
        if not chunked:
            pipe.process(frames, first_frame=rank * F, rows_t=rows_t, want_rows=False, hot_fixed=overlap)
        else:
            exch.k = 0
            packed = mode["packed"]
            if packed:
                # rows of nodes no camera sees are NaN on every rank: they do not travel, and the
                # gather writes the travelling rows packed (row map) straight into the send buffers
                try:
                    exch.set_skipped(engine.skipped_nodes(proj["pix"], want_count=False)[0])
                    pipe.set_row_map(exch.row_map())
                except Exception as ex:      # never exercised on >1 GPU before the scaling run: keep it alive
                    print("bench: packed exchange unavailable (%r), sending every row" % (ex,), file=sys.stderr)
                    mode["packed"] = packed = False
                    exch.set_skipped(None)
                    pipe.set_row_map(None)
            nrows = exch.packed_rows() if packed else N
            for k in range(K):
                c0, fc = exch.my_chunk(k)
                u16 = packed and mode["u16"]
                buf = (chunk_bufs16 if u16 else chunk_bufs)[k][:nrows]
                if fc:
                    try:
                        pipe.process(frames[c0:c0 + fc], first_frame=rank * F + c0, rows_t=buf, want_rows=False,
                                     hot_fixed=overlap)
                    except _capi.UpspError as ex:
                        if not u16:
                            raise
                        print("bench: u16 series refused (%r), sending f32" % (ex,), file=sys.stderr)
                        mode["u16"] = False
                        buf = chunk_bufs[k][:nrows]
                        pipe.process(frames[c0:c0 + fc], first_frame=rank * F + c0, rows_t=buf, want_rows=False,
                                     hot_fixed=overlap)
                exch.submit(buf, packed=packed)
        e[2].record()
        s, ss = pipe.accumulators()
        D.allreduce_sums(s, ss)
        if chunked:
            series = exch.finish()
        avg, rms = pipe.finalize(F * world)
        e[3].record()
        if record:                      # events are read after the timed loop: no host sync inside it
            ev_log.append(e)
            nrays_last[1:] = [proj["pix"]]
        return avg

    for _ in range(a.warmup):
        step(False)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    barrier()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step(True)
    barrier()
    dt = time.perf_counter() - t0
    for e in ev_log:
        t_ray.append(e[0].elapsed_time(e[1]))
        t_frames.append(e[1].elapsed_time(e[2]))
        t_xchg.append(e[2].elapsed_time(e[3]))
    pc = engine.projection_counts(bvh)
    nrays_last[0], primary_rays_last[0], retry_nodes_last[0] = pc["nrays"], pc["primary_rays"], pc["retry_nodes"]
    # the same K steps once more with the library's per-kernel HIP-event timers on
    # (two extra events per launch on the launch stream; kept out of the headline time)
    _capi.timing_enable(True)
    for _ in range(a.steps):
        step(False)
    barrier()
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())

    ms_step = dt / a.steps * 1e3
    total_frames = F * world
    fps = total_frames * a.steps / dt
    ray_ms = float(np.mean(t_ray))
    frm_ms = float(np.mean(t_frames))
    mrays = nrays_last[0] / (ray_ms * 1e-3) / 1e6

    # per-kernel durations: HIP events recorded by the library on the launch stream
    # during the timed steps (upsp_timing_enable / upsp_timing_report)
    timing = _capi.timing_report()
    kernels = {}
    n_retry_rays = 6 * retry_nodes_last[0]
    scene_bytes = bvh.info["device_bytes"]
    gather_launches = -(-F // 64) if not chunked else sum(-(-exch.my_chunk(k)[1] // 64) for k in range(K))
    stream_launches = -(-F // 256) if not chunked else sum(-(-exch.my_chunk(k)[1] // 256) for k in range(K))
    row_launches = -(-F // 1024) if not chunked else sum(-(-exch.my_chunk(k)[1] // 1024) for k in range(K))
    series_rows = exch.packed_rows() if (chunked and mode["packed"]) else N
    series_esz = 2 if (chunked and mode["packed"] and mode["u16"]) else 4
    per_step_bytes = {
        # 40 B per ray (24 B ray + 16 B hit record) + the scene once per launch
        "projection_kernel<primary>": primary_rays_last[0] * 40 + scene_bytes,
        # the occluder-witness pass sees every retry ray (ray + verdict); the traversal that follows
        # only the few it leaves undecided (count known to the device only) plus the scene
        "witness_kernels": n_retry_rays * 40,
        "projection_kernel<retry>": scene_bytes,
        # frame unit = 2 MiB frame + 12 B x N (pix 4, weight 4, out 4).  The 2 MiB
        # compulsory full read of the frame belongs to the hot-pixel scan (the gather's pixel reads
        # hit the Infinity Cache); the gather keeps pix / weight in registers across its 64-frame
        # tile, so per launch it needs 4 B x N x frames written + 8 B x N read once -- counting
        # 12 B x N per FRAME would credit bytes the kernel never has to move.
        # (N > 1: only the rows that travel are stored, as u16 when the values are 16-bit integers)
        "gather_tile_kernel": F * series_esz * series_rows + gather_launches * 8 * N,
        "hot_scan_kernel": F * 2 * size * size,
        # --streamed: pass A reads the frames (its compact buffer stays in cache), pass B writes the series
        "scan_compact_kernel": F * 2 * size * size,
        # (one pass B per 256 frames: index 4 B + flags + accumulators per node and launch)
        "node_stream_kernel": F * series_esz * series_rows + stream_launches * 8 * N,
        # (whole-row pass B: one launch per <= 1024 frames)
        "node_rows_kernel": F * series_esz * series_rows + row_launches * 8 * N,
    }
    for name, (calls, total_ms) in timing.items():
        ms_step_k = total_ms / a.steps
        k = {"calls_per_step": calls / a.steps, "ms_per_step": ms_step_k,
             "avg_launch_ms": total_ms / max(calls, 1)}
        if name in per_step_bytes:
            k["algorithmic_bytes_per_step"] = per_step_bytes[name]
            k["achieved_GBps"] = per_step_bytes[name] / (ms_step_k * 1e-3) / 1e9 if ms_step_k else None
        kernels[name] = k
    dom = max((n for n in kernels if n in per_step_bytes), key=lambda n: kernels[n]["ms_per_step"])
    dk = kernels[dom]
    calls = max(dk["calls_per_step"], 1)
    # HBM traffic of that kernel from the committed rocprofv3 PMC passes (FETCH_SIZE and
    # WRITE_SIZE collected separately; FETCH_SIZE doubled per the gfx950 correction of
    # MI355X_MICROARCH.md "HBM"; counter unit = KB) -- per launch, same launch shape
    traffic = None
    prof = os.path.join(ROOT, "profiles", "r01_bench_summary.json")
    pkey = {"gather_tile_kernel": "gather_tile16_kernel<4, true, false>", "hot_scan_kernel": "hot_scan_kernel",
            "scan_compact_kernel": "scan_compact_kernel<true>", "node_stream_kernel": "node_stream_kernel",
            "projection_kernel<primary>": "projection_kernel<false, 0>",
            "projection_kernel<retry>": "projection_kernel<false, 1>"}.get(dom)
    if os.path.exists(prof) and pkey:
        pj = json.load(open(prof)).get(pkey, {})
        if "FETCH_SIZE_KB_per_launch" in pj and "WRITE_SIZE_KB_per_launch" in pj:
            traffic = (2 * pj["FETCH_SIZE_KB_per_launch"] + pj["WRITE_SIZE_KB_per_launch"]) * 1024
    roof = {"kernel": dom, "bound": "hbm",
            "achieved": dk["achieved_GBps"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": dk["achieved_GBps"] / HBM_PEAK_GBS, "traffic": traffic,
            "algorithmic_bytes_per_launch": per_step_bytes[dom] / calls,
            "avg_launch_ms": dk["avg_launch_ms"], "launches_per_step": calls}

    out = {
        "metric": "frames/s", "value": fps, "unit": "frames/s", "n_gpus": world,
        "steps": a.steps, "warmup": a.warmup, "ms_per_step": ms_step, "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "f32",
        "data": "synthetic", **({"backend": backend} if world > 1 and backend != "nccl" else {}),
        "config": {"workload": "configs[%d]: %d frames/GPU x %dx%d u16, %d-tri tunnel model (%d nodes), "
                               "raycast+%sprojection" % (2 if a.registration else 1, F, size, size,
                                                         tris.shape[0], N,
                                                         "registration+" if a.registration else ""),
                   "frames_per_gpu": F, "nodes": N, "triangles": int(tris.shape[0]),
                   "parallelism": "frames sharded x%d" % world,
                   "schedule": ("hot-pixel scan of all frames on a side stream, concurrent with the projection build"
                                if overlap else "projection build, then scan + gather kernels per 64-frame sub-batch"
                                if (a.two_kernel or a.registration) else
                                "projection build, then scan + compact / node stream passes per 64-frame sub-batch"),
                   **({"exchange": "%d chunks, %s rows as %s" % (K, "visible" if mode["packed"] else "all",
                                                                 "u16" if series_esz == 2 else "f32")}
                      if chunked else {})},
        "mrays_per_s": mrays, "rays_per_step": nrays_last[0],
        "rays_cast_per_step": primary_rays_last[0] + 6 * retry_nodes_last[0],
        "breakdown_ms": {"projection_build": ray_ms, "frame_loop": frm_ms,
                         "exchange_finals": float(np.mean(t_xchg))},
        "frame_loop_frames_per_s": F / (frm_ms * 1e-3),
        # whole frame loop against HBM: (2 MiB + 4 B x N) per frame + 8 B x N per 64-frame tile
        "frame_loop_GBps": (F * (2 * size * size + series_esz * series_rows) + gather_launches * 8 * N) / (frm_ms * 1e-3) / 1e9,
        "roofline": roof,
        "kernels": kernels,
    }
    if world == 1 and not a.registration and not a.no_reraycast:
        # SURVEY.md 8(d) stress mode "frame with re-raycast": one projection build (N_nodes visibility
        # rays + retries) per frame instead of per run (docs/sphinx/known-issues.rst:18-30: model motion)
        nrr = 20
        one = torch.empty((N, engine.series_ld(1)), dtype=torch.float32, device="cuda")[:, :1]
        pipe.reset()
        torch.cuda.synchronize()
        r0, r1 = ev(), ev()
        r0.record()
        for i in range(nrr):
            pr = engine.build_projection(bvh, cam, d_nodes, d_nrm, d_tn, 70.0, counts=False)
            pipe.set_projection(0, pr["pix"])
            pipe.process(frames[i:i + 1], first_frame=i, rows_t=one, want_rows=False)
        r1.record()
        torch.cuda.synchronize()
        out["reraycast_frames_per_s"] = nrr / (r0.elapsed_time(r1) * 1e-3)
    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(verts, tris, cd, size, F, nrays_last[1].cpu().numpy())
    elif rank == 0:
        out["cpu_baseline"] = None
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
