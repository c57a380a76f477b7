#!/usr/bin/env python3
"""bench.py -- BASELINE.json metric on MI355X.

One STEP = one pass of the hot path over one batch of synthetic input
(BASELINE.json configs[1]: 1 000 frames, 1024x1024, ~1 M-triangle wind-tunnel model,
ray cast + projection only):

    projection build  (create_projection_mat: ~0.5 M node rays + jitter retries)
  + frame loop        (hot-pixel repair -> nearest-pixel projection -> double
                       accumulators -> node-major time series) over every frame
  + finals            (avg / rms)
  + for N > 1: the end-of-run exchanges (all-reduce of the accumulators and the
    time-series exchange, grouped point-to-point sends over RCCL/xGMI: upsp_exchange_* of libupsp_gpu.so)

with mesh, BVH, camera and all frames already resident in HBM.  Frames shard over
ranks (weak scaling: every GPU processes --frames frames).

`python bench.py --gpus N` with N > 1 and no rank environment starts N rank processes itself
(torch.distributed.run, one per GPU, RCCL) before anything touches the GPU; under an external
launcher (RANK / WORLD_SIZE set) it is one of the ranks.  Rank 0 prints ONE JSON line.
`value` = frames/s over the whole job (`summary` says it in one sentence, with the host-feed rate beside it).
`mrays_per_s` = closest-hit rays that ENTER the tree per second, one ray per pixel (`pixel_rays_fill`: a 1 M-triangle
sphere that fills the frame; `pixel_rays`: the tunnel model, 8 % of the pixels), every ray traversed, a sample checked
against the oracle, with the traversal bytes per ray (wide nodes visited x 128 B + triangles tested x 48 B)
from the statistics counters.  The projection build applies the oblique test before the rays (nodes it rejects have no
entry whatever their rays say and cast none) and decides most retry rays by the occluder witness instead of a traversal:
`rays_cast_per_step` counts what is really cast (`mrays_cast_per_s` over the build time), `rays_per_step` what the
REFERENCE casts for the same camera, and `mrays_reference_equivalent_per_s` is that count over the build time -- an
equivalence, not a ray rate.

Schedule of the default (one GPU, plain loop): the ray casting of the projection build runs on a high-priority stream of its own
(it depends on nothing the frame loop does, so it does not wait for the previous step's pass B either: the device starts it when
the previous build has left that stream), pass A of the frame loop (which needs only the candidate pixels of the in-frame nodes)
beside it, then pass B and the repair; `--serial` = one stream, stage after stage.  N > 1 (`--gpus N`, or `--force-chunked` on one
GPU): the same arrangement -- pass A once for the rank's frames beside the build --, then the active pixels' u16 series travel, one
block per peer (`--row-wire`: packed node rows; `--wire12`: packed to 12 bits; `--chunk-scan`: pass A per chunk, every chunk's sends
beside the next chunk's scan), through the library's exchange over RCCL, and the owner of a node runs pass B; between GPUs two
exchanges are used in turn (a step's blocks are on the links during the next step; the run's last exchange is finished before the
clock stops).  No cyclic-GC pass runs inside a timed loop (`quiet_gc`).

`configs2` (default run) / `--registration`: BASELINE configs[2], per-frame ECC registration in front of the projection,
with its own roofline (ecc_sums_kernel), iterations per frame, CPU baseline (`register_pixel` included) and parity block
(warps, iteration counts, warped rows against the oracle on the frames the CPU baseline registered).
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)


_PEAK = {}


def measured_peak():
    """the roofline's second denominator, measured in THIS process by HIP events -- a read-only and a write-only
    stream over 1 GiB in the access shapes of the frame loop's two passes (upsp_bandwidth_probe: non-temporal 16-byte loads,
    several in flight per lane / a workgroup sweeping whole 4-KB row pieces with 16-byte non-temporal stores; the fastest of four
    launch shapes each).  (Until round 5 a float4 COPY kernel was the denominator: it mixes the two streams and is slower than
    either pass -- 5.3 TB/s against 6.0-6.6 for pass A -- so fractions came out above 1.)"""
    if "v" not in _PEAK:
        from upsp_processing_amd import _capi
        _PEAK["v"] = _capi.bandwidth_probe(1 << 30, 5)
    return _PEAK["v"]


# which stream a kernel's algorithmic bytes are: fraction READ (the rest written)
READ_SHARE = {"scan_compact_kernel": 1.0, "hot_scan_kernel": 1.0, "ecc_sums_kernel": 1.0, "ecc_sums_identity": 1.0, "ecc_sums_general": 1.0,
              "node_rows_kernel": 0.0, "node_rows_multi_kernel": 0.0, "gather_tile_kernel": 0.0,
              "gauss_pass_kernels": 1.0 / 3.0, "ecc_blur_ident_kernel": 0.6, "warp_u16_kernel": 12.0 / 14.0, "scan_compact_multi": 1.0}


def floor_ms(name, nbytes, pk):
    """time the measured read / write rates allow for `nbytes` algorithmic bytes of kernel `name`"""
    r = READ_SHARE.get(name, 0.5)
    return (nbytes * r / pk["read_GBps"] + nbytes * (1.0 - r) / pk["write_GBps"]) / 1e9 * 1e3


def roofline_extras(roof, bytes_by_kernel, ms_step):
    """peak_measured / frac_of_measured (the dominant kernel against the measured rate of ITS stream: read-only for pass A and the
    ECC sums, write-only for pass B) and step_frac (ALL algorithmic bytes of a step over the whole step time, against the spec peak;
    step_frac_of_measured: the time the measured rates allow for every kernel's bytes over the step time)."""
    pk = measured_peak()
    dom = roof["kernel"]
    r = READ_SHARE.get(dom, 0.5)
    kind = "read-only" if r == 1.0 else "write-only" if r == 0.0 else "read share %.2f" % r
    roof["peak_measured"] = 1.0 / (r / pk["read_GBps"] + (1.0 - r) / pk["write_GBps"])
    roof["peak_measured_kind"] = ("%s stream in the kernel's own access shape, in this process (upsp_bandwidth_probe, %d MiB, %d launches, HIP "
                                  "events): read-only %.0f GB/s, write-only %.0f GB/s" % (kind, pk["bytes"] >> 20, pk["reps"], pk["read_GBps"], pk["write_GBps"]))
    roof["frac_of_measured"] = roof["achieved"] / roof["peak_measured"]
    step_bytes = sum(bytes_by_kernel.values())
    roof["step_algorithmic_bytes"] = int(step_bytes)
    roof["step_frac"] = step_bytes / (ms_step * 1e-3) / 1e9 / HBM_PEAK_GBS
    roof["step_frac_of_measured"] = sum(floor_ms(k, v, pk) for k, v in bytes_by_kernel.items()) / ms_step
    return roof


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
    ap.add_argument("--row-wire", action="store_true",
                    help="N>1 / --force-chunked: the node ROWS travel (packed u16 rows of the nodes some camera sees).  Default: the "
                         "series of the ACTIVE PIXELS travel (a third of the bytes on the bench model: several nodes read one pixel) "
                         "and the owner of a node runs pass B over all frames of the run")
    ap.add_argument("--wire12", action="store_true",
                    help="N>1 (RCCL): the u16 series packed to 12 bits for the wire (3 bytes per 2 frames; 12-bit cameras)")
    ap.add_argument("--two-kernel", action="store_true",
                    help="frame loop as scan kernel + gather kernel per 64 frames instead of the (default) "
                         "streamed two-pass schedule (pass A: scan + compact pixel series, pass B: whole rows)")
    ap.add_argument("--model", default="quad", choices=["quad", "uv", "5m"],
                    help="quad: cube-sphere tunnel model (valence <= 6); uv: UV-sphere model with "
                         "1000-valent polar fans (worst case for per-ray traversal length); 5m: the 5 M-triangle "
                         "tunnel model of configs[4] (use with --cameras)")
    ap.add_argument("--cameras", type=int, default=1,
                    help="configs[4] shape: N cameras around the model (azimuth 0, 90, 180, 270 ...), AverageViews weights, "
                         "one frame SET (a frame of every camera) per step unit; on one GPU")
    ap.add_argument("--fill-frame", action="store_true",
                    help="1 M-triangle sphere filling the frame instead of the tunnel model: ~0.2 M visible nodes on "
                         "as many active pixels, so the compact pixel series (2 KB per active pixel and 1000 frames) "
                         "no longer fit the Infinity Cache")
    ap.add_argument("--overlap", action="store_true",
                    help="(the default since round 3; accepted for old command lines) pass A of the frame loop (hot-pixel count "
                         "+ compact pixel series, from the candidate pixels of the in-frame nodes) on a second stream while the "
                         "rays of the projection build are cast")
    ap.add_argument("--chunks", type=int, default=0,
                    help="N > 1 loop: chunks the rank's frames are exchanged in (the exchange of chunk k runs while chunk k + 1 is scanned); "
                         "0 = 1 (pass A once beside the build, one block per peer), or 4 with --chunk-scan / the row wire")
    ap.add_argument("--config3-share", action="store_true",
                    help="one rank's share of BASELINE configs[3] (100 000 frames on 8 GPUs) as ONE step: 12 500 resident frames through "
                         "the N > 1 loop (implies --force-chunked on one GPU; as many exchange chunks as keep each within one pass A, "
                         "i.e. <= 1024 frames) -- the run's chunks overlap each other, only the last one and pass B are exposed once "
                         "per 12 500 frames, not once per 1000 as in the default N > 1 step")
    ap.add_argument("--chunk-scan", action="store_true",
                    help="N > 1 loop: pass A per chunk after the projection build (default: once for all frames of the rank, beside the build)")
    ap.add_argument("--defer-exchange", action="store_true",
                    help="--force-chunked on one GPU: the two-exchanges-in-turn schedule of the N > 1 runs (see --sync-exchange)")
    ap.add_argument("--sync-exchange", action="store_true",
                    help="N > 1 loop: finish every step's exchange (wait for its last chunk, run the owner's pass B) inside the step, "
                         "as round 3 did.  Default between GPUs: two exchanges in turn -- step k's series are placed and its pass B runs "
                         "after step k + 1's chunks are on their way, so a step waits for no link; the last step's are finished before "
                         "the clock stops")
    ap.add_argument("--serial", action="store_true",
                    help="one stream: projection build, then pass A, then pass B (round 2's default schedule)")
    ap.add_argument("--plain-frames", action="store_true",
                    help="round-1 frame content (no background, fiducial discs or hot pixels)")
    return ap.parse_args()


def spawn_ranks(n):
    """`--gpus N` without a rank environment: start the N ranks as fresh child processes (nothing in
    this process has touched the GPU) and pass rank 0's line through."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd)


class quiet_gc:
    """No cyclic-GC pass inside a timed loop: with torch imported a full collection walks ~1 M objects and stops the issuing
    thread for 40-65 ms -- measured as ONE 32-61-ms step in the first bench process of a fresh box (N > 1 loop, one GPU), where
    the host is not a step ahead of the GPU.  Collected before, frozen, re-enabled after."""

    def __enter__(self):
        import gc
        self.was = gc.isenabled()
        gc.collect()
        gc.freeze()
        gc.disable()

    def __exit__(self, *exc):
        import gc
        if self.was:
            gc.enable()
        gc.unfreeze()
        return False


def hot_pixel_restorer(frames, thresh=4064):
    """The frame loop repairs hot pixels IN PLACE (like the reference): after the first step no frame would hold one and the
    repair / re-projection branch would never run inside the timed region.  fix_hot_pixels writes only pixels >= thresh
    (cv_extras.cpp:249-274), so putting THOSE pixels back before every step -- a scatter of a few dozen values, inside the timed
    region -- makes every step see the frames as they arrived.  (Until round 4 the whole frames were copied back: 10 x 2 MiB
    through an index_put kernel, 33 us of a 1-ms step that belonged to the bench, not to the path.)  Returns restore()."""
    import torch
    F = frames.shape[0]
    f16 = frames.view(torch.int16)
    npx = f16[0].numel()
    pos, val = [], []
    for f0 in range(0, F, 500):                     # (in pieces: a 12 500-frame share is 13 G pixels)
        sub = f16[f0:f0 + 500].reshape(min(500, F - f0), -1)
        fi, pi = torch.nonzero(sub >= thresh, as_tuple=True)
        pos.append((fi + f0) * npx + pi)
        val.append(sub[fi, pi])
    pos, val = torch.cat(pos), torch.cat(val).clone()
    flat = f16.reshape(-1)

    def restore():
        if pos.numel():
            flat[pos] = val
    return restore


def usable_cpus():
    """Host cores this process may really use: the affinity mask and the cgroup CPU quota (a one-GPU box
    of the pool exposes all of the host's logical CPUs but grants a share of them)."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(round(int(txt[0]) / int(txt[1])))))
            else:
                q = int(txt[0])
                per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                if q > 0:
                    n = min(n, max(1, int(round(q / per))))
            break
        except (OSError, ValueError, IndexError):
            continue
    return n


def pixel_rays(cam_dict, size):
    """One ray per pixel of the size x size frame: camera centre -> through the pixel (pinhole part of the
    calibration; the bench camera has no distortion).  Returns (origin f32[3], dirs f32[size*size, 3])."""
    K, R, t = [np.asarray(cam_dict[k], np.float64) for k in ("K", "R", "t")]
    c = -R.T @ t
    v, u = np.meshgrid(np.arange(size, dtype=np.float64), np.arange(size, dtype=np.float64), indexing="ij")
    pc = np.stack([(u - K[0, 2]) / K[0, 0], (v - K[1, 2]) / K[1, 1], np.ones_like(u)], -1).reshape(-1, 3)
    d = pc @ R                                  # R^T applied to every row
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    return c.astype(np.float32), d.astype(np.float32)


def orc_bvh(tris9):
    from oracle import oracle as orc
    return orc.OracleBVH(tris9)


def rays_entering(org, dirs, lo, hi):
    """Rays whose LINE pierces the box [lo, hi] (slab test in double on the host; the count reported beside the
    traversal rate -- the traversal itself uses the library's own box test)."""
    o = np.asarray(org, np.float64).reshape(1, 3)
    d = np.asarray(dirs, np.float64)
    with np.errstate(divide="ignore", invalid="ignore"):
        t0, t1 = (np.asarray(lo, np.float64) - o) / d, (np.asarray(hi, np.float64) - o) / d
    tn, tf = np.fmin(t0, t1).max(1), np.fmax(t0, t1).min(1)
    return int((tn <= tf).sum())


def pixel_ray_rate(bvh, cam_dict, size, check_with=None):
    """The other half of BASELINE's metric taken literally: closest hit of ONE RAY PER PIXEL of the 1 Mpix frame
    against the 1 M-triangle model through upsp_bvh_intersect (rt::BVH::intersect semantics: t, primID) -- every ray
    is traversed, nothing is culled or witnessed.  check_with: oracle BVH -> a strided sample is compared bit for bit.
    Reported: all rays / s, the rays that enter the root box / s (the others end at the first box test), the hit
    fraction, and SURVEY.md 8(d)'s traversal bytes per ray from the statistics counters of one extra, untimed call
    (wide nodes visited x 128 B -- one record decides two levels of rt::BVH::intersect's descent, pspRT.cpp:376-429 --
    + triangles tested x 48 B)."""
    import torch
    org, dirs = pixel_rays(cam_dict, size)
    d_org, d_dirs = torch.as_tensor(org).cuda(), torch.as_tensor(dirs).cuda()
    for _ in range(2):
        h = bvh.intersect(d_org, d_dirs, want=("hit", "t", "prim"))
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 5
    e0.record()
    for _ in range(reps):
        h = bvh.intersect(d_org, d_dirs, want=("hit", "t", "prim"))
    e1.record()
    torch.cuda.synchronize()
    bvh.check()
    ms = e0.elapsed_time(e1) / reps
    n = dirs.shape[0]
    info = bvh.info
    entered = rays_entering(org, dirs, info["bounds_min"], info["bounds_max"])
    bvh.enable_stats(True)                      # one-lane traversal of every ray with the counters on (not timed)
    bvh.intersect(d_org, d_dirs, want=("hit",))
    st = bvh.last_stats()
    bvh.enable_stats(False)
    trav_bytes = st["nodes"] * 128 + st["tris"] * 48
    alg = n * 40                                 # 24 B ray + 16 B hit record
    out = {"rays": int(n), "ms": ms, "mrays_per_s": n / (ms * 1e-3) / 1e6,
           "rays_entered": entered, "mrays_entered_per_s": entered / (ms * 1e-3) / 1e6,
           "hit_fraction": float(h["hit"].float().mean().item()),
           "nodes_visited_per_entered_ray": st["nodes"] / max(entered, 1),
           "tris_tested_per_entered_ray": st["tris"] / max(entered, 1),
           "traversal_bytes_per_ray": trav_bytes / max(n, 1),
           "traversal_bytes_per_entered_ray": trav_bytes / max(entered, 1),
           "traversal_GBps": trav_bytes / (ms * 1e-3) / 1e9,
           "algorithmic_bytes": int(alg), "achieved_GBps": alg / (ms * 1e-3) / 1e9, "frac_of_hbm_peak": alg / (ms * 1e-3) / 8e12,
           "scene_bytes": int(info["device_bytes"]),
           "what": "closest hit (t, primID) of one ray per pixel of the %d x %d frame, upsp_bvh_intersect; all rays traversed; "
                   "algorithmic bytes = 40 B per ray (the scene, %d MB, is cache-resident across the batch and not credited); "
                   "traversal bytes = wide nodes visited x 128 B + triangles tested x 48 B from the statistics counters"
                   % (size, size, info["device_bytes"] // 1000000)}
    if check_with is not None:
        idx = np.arange(0, dirs.shape[0], 37)
        t0 = time.perf_counter()
        o = check_with.intersect(org, dirs[idx], threads=usable_cpus())
        dt = time.perf_counter() - t0
        g_hit, g_t, g_prim = [h[k].cpu().numpy()[idx] for k in ("hit", "t", "prim")]
        out["parity_sample"] = int(idx.size)
        out["parity"] = bool(np.array_equal(g_hit, o["hit"]) and np.array_equal(g_prim, o["prim"])
                             and np.array_equal(g_t.view(np.int32), o["t"].view(np.int32)))
        out["cpu_mrays_per_s"] = idx.size / dt / 1e6
    return out


def cpu_baseline(verts, tris, cam_dict, size, nframes_step, sample, registration=False):
    """Oracle (CPU restatement, kind 'port') on a bounded sample of the same workload: the projection
    build on the full model (OpenMP over node blocks like psp_process.cpp:218-260) and the frame loop
    (OpenMP over frames like psp_process.cpp:1742-1851) on the first `len(sample)` frames of the step."""
    from oracle import oracle as orc
    from upsp_processing_amd import synthetic as syn, engine
    cores = usable_cpus()
    s9, tn = syn.soup(verts, tris)
    nrm = syn.node_normals(verts, tris)
    t0 = time.perf_counter()
    obv = orc.OracleBVH(s9)
    t_build = time.perf_counter() - t0
    cam = orc.make_camera(cam_dict["K"], cam_dict["dist"], cam_dict["R"], cam_dict["t"], size, size)
    t0 = time.perf_counter()
    r = orc.create_projection(obv, cam, verts, nrm, tn, engine.oblique_threshold(70.0), threads=cores)
    t_proj = time.perf_counter() - t0
    mrays = r["nrays"] / t_proj / 1e6
    pix = np.ascontiguousarray(r["pix"], dtype=np.int32)
    frames = sample.copy()
    tm = {}
    t0 = time.perf_counter()
    _, s, ss = orc.frame_loop(frames, pix, want_rows=False, threads=cores, timing=tm)
    t_frames = time.perf_counter() - t0
    # per frame: the loop proper; per run: every thread allocates / first-touches and later merges its
    # own 2 x N doubles (psp_process.cpp:1744-1745, 1845-1850), amortised over a whole run
    per_frame = tm["loop"] / frames.shape[0]
    t_fixed = tm["setup"] + tm["merge"]
    reg_note, reg = "", None
    if registration:
        # configs[2]: register_pixel (ECC + warp, cpp/lib/registration.cpp:32-81) per frame before the projection,
        # after fix_hot_pixels like the loop (psp_process.cpp:1771-1795); one frame per thread like the reference's
        # OpenMP loop, bounded sample.  The template is the RAW first frame (psp_process.cpp:2057-2058).
        from concurrent.futures import ThreadPoolExecutor
        nreg = min(sample.shape[0] - 1, 2 * cores)
        ref32 = sample[0].astype(np.float32)

        def one(f):
            img, _ = orc.fix_hot_pixels(sample[f])
            out, M, it = orc.register_pixel(ref32, img)
            return img, M, it, orc.project_frame(out, pix, None)
        t0 = time.perf_counter()
        with ThreadPoolExecutor(cores) as ex:
            res = list(ex.map(one, range(1, nreg + 1)))
        t_reg = time.perf_counter() - t0
        per_frame += t_reg / nreg
        reg = dict(first=1, fixed=[r[0] for r in res], M=np.stack([r[1] for r in res]), its=np.array([r[2] for r in res]),
                   rows=[r[3] for r in res], seconds_per_frame=t_reg / nreg * cores)
        reg_note = " + fix_hot_pixels, register_pixel, project_frame on %d frames (%.2f s, %.1f ECC iterations per frame)" % (
            nreg, t_reg, float(np.mean(reg["its"])))
    # a few rows for the parity check of the series (single thread, rows kept)
    few = sample[:8].copy()
    rows8, _, _ = orc.frame_loop(few, pix, want_rows=True, threads=1)
    fps = nframes_step / (t_proj + t_fixed + per_frame * nframes_step)
    out = {"value": fps, "unit": "frames/s", "cores": cores, "kind": "port",
           "sample": "oracle/ (C, OpenMP, %d threads): projection build on the full model (%d rays, %.2f s) "
                     "+ %d frames of the frame loop (%.3f s: %.3f ms per frame + %.2f s per run for the "
                     "thread-private accumulators)%s; extrapolated to the %d-frame step"
                     % (cores, r["nrays"], t_proj, frames.shape[0], t_frames, tm["loop"] / frames.shape[0] * 1e3, t_fixed,
                        reg_note, nframes_step),
           "mrays_per_s": mrays, "frame_loop_frames_per_s": 1.0 / per_frame, "bvh_build_s": t_build}
    return out, dict(pix=pix, nrays=int(r["nrays"]), sum=s, sumsq=ss, rows8=rows8, frames_fixed=frames, obv=obv, reg=reg,
                     t_proj=t_proj, t_fixed=t_fixed, loop_per_frame=tm["loop"] / frames.shape[0], cores=cores)


def registration_parity(ref, sample, gpix_dev, N, size):
    """configs[2] parity of THIS run: the frames the CPU baseline registered (oracle: fix_hot_pixels -> register_pixel
    -> project_frame) through a fresh registration pipeline on the GPU.  Bars (tests/test_imageops_gpu.py): warp
    matrix max|dM| <= 1e-4 (linear part) / 2e-3 px (translation), identical iteration counts, series rows bit-exact
    against project_frame(warpAffine(frame, M_gpu)) (exact integer arithmetic), and against the oracle's own chain
    (its M) within ONE STEP of the warp's fixed-point coordinates: warpAffine quantises source positions to 1/32 px, so
    two matrices that differ by 1e-5 px put a few pixels one step apart in x and / or y -- an intensity difference of at
    most 2 x (largest step between neighbouring pixels of the frame) / 32, + 1 count of rounding."""
    import torch
    from oracle import oracle as orc
    from upsp_processing_amd import engine
    reg = ref["reg"]
    nreg = len(reg["fixed"])
    d = torch.as_tensor(sample[:nreg + 1].view(np.int16)).view(torch.uint16).cuda()
    p = engine.FramePipeline(1, size, size, N, registration=1)
    p.set_projection(0, gpix_dev)
    p.set_reference(0, d[0].to(torch.float32))
    warps = torch.zeros((nreg + 1, 1, 6), dtype=torch.float32, device="cuda")
    iters = torch.zeros((nreg + 1, 1), dtype=torch.int32, device="cuda")
    rows = p.process(d, 0, warps=warps, ecc_iters=iters).cpu().numpy()
    w = warps.cpu().numpy()[1:, 0].reshape(nreg, 2, 3)
    it = iters.cpu().numpy()[1:, 0]
    pix = ref["pix"]
    ok = pix >= 0
    d_lin = float(np.abs(w[:, :, :2] - reg["M"][:, :, :2]).max())
    d_tr = float(np.abs(w[:, :, 2] - reg["M"][:, :, 2]).max())
    exact, worst, within = True, 0.0, True
    for i in range(nreg):
        fx = reg["fixed"][i].astype(np.int32)
        want = orc.project_frame(orc.warp_affine(reg["fixed"][i], w[i], 1), pix, None)
        exact = exact and bool(np.array_equal(rows[i + 1, ok].view(np.int32), want[ok].view(np.int32)))
        step = max(int(np.abs(np.diff(fx, axis=0)).max()), int(np.abs(np.diff(fx, axis=1)).max()))
        dI = float(np.abs(rows[i + 1, ok] - reg["rows"][i][ok]).max())
        within = within and dI <= 2.0 * step / 32.0 + 1.0
        worst = max(worst, dI)
    checks = {
        "ecc_warp_linear_1e-4": d_lin <= 1e-4, "ecc_warp_translation_2e-3_px": d_tr <= 2e-3,
        "ecc_iteration_counts": bool(np.array_equal(it, reg["its"])),
        "rows_bitexact_for_gpu_warp": exact, "rows_vs_oracle_chain_one_warp_step": within,
    }
    return checks, {"frames": nreg, "max_dM_linear": d_lin, "max_dM_translation_px": d_tr, "max_dI_vs_oracle_chain": worst,
                    "iterations": [int(x) for x in it]}


def registration_block(bvh, cam, d_nodes, d_nrm, d_tn, frames, restore, N, size, n_active, steps, warmup):
    """BASELINE configs[2] on the resident frames: projection build + (fix_hot_pixels -> ECC registration -> warp of the
    pixels nodes read -> projection) over every frame + finals.  Returns the sub-block of the JSON line."""
    import torch
    from upsp_processing_amd import _capi, engine
    F = frames.shape[0]
    npx = size * size
    pipe = engine.FramePipeline(1, size, size, N, registration=1)
    restore()
    pipe.set_reference(0, frames[0].to(torch.float32))      # raw first frame as ECC template (psp_process.cpp:2057)
    # (registration as the last image stage: ONE whole-row pass B per <= 1024 frames -> the plain multiple of 256 B as pitch)
    rows_t = torch.empty((N, engine.series_ld(F, whole_rows=True)), dtype=torch.float32, device="cuda")[:, :F]
    pipe.set_row_padding(True)    # (columns F .. pitch of rows_t are padding)

    side = torch.cuda.Stream(priority=-1)       # the build of a step runs beside the previous step's registration (see main())
    side.wait_stream(torch.cuda.current_stream())   # (once: whatever the caller still has in flight on the model's arrays)

    def step():
        restore()
        main = torch.cuda.current_stream()
        with torch.cuda.stream(side):
            proj = engine.build_projection(bvh, cam, d_nodes, d_nrm, d_tn, 70.0, counts=False)
        main.wait_stream(side)
        proj["pix"].record_stream(main)
        pipe.reset(deferred=True)
        pipe.set_projection(0, proj["pix"])
        pipe.process(frames, first_frame=0, rows_t=rows_t, want_rows=False)
        return pipe.finalize(F)

    for _ in range(warmup):
        step()
    torch.cuda.synchronize()
    st0 = pipe.ecc_stats()
    with quiet_gc():
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    st1 = pipe.ecc_stats()
    iters = (st1["frame_iterations"] - st0["frame_iterations"]) / max(st1["frames"] - st0["frames"], 1)
    _capi.timing_enable(True)
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    _capi.timing_enable(False)
    rep = merge_ecc_labels(_capi.timing_report(spread=True))
    fused = "ecc_blur_ident_kernel" in rep and "ecc_sums_identity" not in rep
    ms_step = dt / steps * 1e3
    bytes_step = {
        # SURVEY.md 8(d) with registration: 8 B per pixel and ECC iteration (blurred frame + template, gradients
        # recomputed on the fly); pre-blur 2 B in + 4 B out per pixel; the warp produces only the pixels a node reads:
        # per listed pixel 4 B list entry + 4 x 2 B source pixels + 2 B out
        "ecc_sums_kernel": iters * F * 8 * npx,
        # the two kernels behind that label: every frame's FIRST iteration starts from the identity warp
        # (ecc_cols_kernel<true,..>), the others run the general kernel (ecc_cols_kernel<false,..>)
        "ecc_sums_identity": F * 8 * npx,
        "ecc_sums_general": max(iters - 1.0, 0.0) * F * 8 * npx,
        "gauss_pass_kernels": F * 6 * npx,
        # round 6: the pre-blur and the identity iteration in one pass (ecc_blur_ident_kernel): 2 B frame in + 4 B blurred frame out
        # + 4 B template per pixel -- 10 B where the two kernels move 6 + 8
        "ecc_blur_ident_kernel": F * 10 * npx,
        "warp_u16_kernel": F * n_active * 14,
        "hot_scan_kernel": F * 2 * npx,
        "gather_tile_kernel": F * 4 * N + (-(-F // 64)) * 8 * N,
        "node_rows_kernel": F * 4 * N + (-(-F // 1024)) * 8 * N,
    }
    if fused:       # the sums launches that remain are the general iterations
        bytes_step["ecc_sums_kernel"] = bytes_step["ecc_sums_general"]
    kernels = {}
    for name, (calls, total, lo, med, hi) in rep.items():
        k = {"calls_per_step": calls / steps, "ms_per_step": total / steps, "avg_launch_ms": total / max(calls, 1)}
        if name in bytes_step and total:
            k["algorithmic_bytes_per_step"] = bytes_step[name]
            k["achieved_GBps"] = bytes_step[name] / (total / steps * 1e-3) / 1e9
        kernels[name] = k
    # the dominant KERNEL (the blend "ecc_sums_kernel" of the two sums kernels is reported beside it, not as the roofline)
    dom = max((n for n in kernels if "achieved_GBps" in kernels[n] and n != "ecc_sums_kernel"), key=lambda n: kernels[n]["ms_per_step"])
    dk = kernels[dom]
    ecc_fracs = None
    if "ecc_sums_kernel" in kernels:
        # SURVEY 8(d) credits 8 B per pixel and frame-iteration (blurred frame + template); one LAUNCH reads the 4-MiB template
        # once for all of its frames (it stays in cache), so the bytes a launch must move are 4 B per pixel and frame-iteration
        # + 4 B per pixel and launch: both fractions are reported
        ek = kernels["ecc_sums_kernel"]
        once = (iters - 1.0 if fused else iters) * F * 4 * npx + ek["calls_per_step"] * 4 * npx
        ecc_fracs = {"survey_8B_per_px_iteration": ek["achieved_GBps"] / HBM_PEAK_GBS,
                     "template_once_per_launch": once / (ek["ms_per_step"] * 1e-3) / 1e9 / HBM_PEAK_GBS,
                     "identity_launch_ms": kernels.get("ecc_sums_identity", {}).get("avg_launch_ms"),
                     "blur_ident_launch_ms": kernels.get("ecc_blur_ident_kernel", {}).get("avg_launch_ms"),
                     "general_launch_ms": kernels.get("ecc_sums_general", {}).get("avg_launch_ms")}
    return {
        "ecc_sums_fraction_of_hbm_peak": ecc_fracs,
        "workload": "configs[2]: %d frames x %dx%d u16, per-frame ECC registration + projection, projection build per step" % (F, size, size),
        "value": F * steps / dt, "unit": "frames/s", "frames": F, "steps": steps, "warmup": warmup, "ms_per_step": ms_step,
        "ecc_iterations_per_frame": iters,
        "roofline": roofline_extras(dict({"kernel": dom, "bound": "hbm", "achieved": dk["achieved_GBps"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
                          "frac": dk["achieved_GBps"] / HBM_PEAK_GBS, "traffic": tracked_traffic(dom, "ecc")[0],
                          "traffic_source": tracked_traffic(dom, "ecc")[1],
                          "algorithmic_bytes_per_launch": dk["algorithmic_bytes_per_step"] / max(dk["calls_per_step"], 1),
                          "avg_launch_ms": dk["avg_launch_ms"], "launches_per_step": dk["calls_per_step"]},
                         **ECC_SYMBOLS.get(dom, {})),
                         # a step's algorithmic bytes: sums 8 B / px / iteration, pre-blur 6 B / px (fused: 10 B / px for the pre-blur
                         # with the first iteration), warp 14 B / active px, pass B rows
                         {k: bytes_step[k] for k in ("ecc_sums_kernel", "ecc_blur_ident_kernel" if fused else "gauss_pass_kernels",
                                                     "warp_u16_kernel", "node_rows_kernel")}, ms_step),
        "kernels": kernels,
    }


# "ecc_sums_kernel" = the ECC sums launches (the library times them as ecc_sums_identity / ecc_sums_general); the symbols a
# rocprofv3 trace shows for them
ECC_SYMBOLS = {"ecc_sums_kernel": {"kernel_symbols": ["ecc_cols_kernel<true,4,5,true> (iterations from the identity warp, when not fused with the pre-blur)",
                                                      "ecc_cols_kernel<false,2,4,true> (general warp: source taps from an LDS tile)"]},
               "ecc_sums_identity": {"kernel_symbols": ["ecc_cols_kernel<true,4,5,true> (a frame's first iteration: identity warp)"]},
               "ecc_sums_general": {"kernel_symbols": ["ecc_cols_kernel<false,2,4,true> (general warp: source taps from an LDS tile)"]},
               "ecc_blur_ident_kernel": {"kernel_symbols": ["ecc_blur_ident_kernel<true,4> (5 x 5 pre-blur + the first iteration's sums in one pass)"]}}


def tracked_traffic(kernel, kind, world=1):
    """HBM traffic of a kernel per launch: only from a rocprofv3 PMC summary of THIS configuration (tools/profile_bench.sh writes
    it: FETCH_SIZE and WRITE_SIZE in separate passes, the gfx950 FETCH correction applied for the streaming kernels) -- the file
    UPSP_BENCH_TRAFFIC_JSON names, else the newest tracked profiles/rNN_<kind>_summary.json, used only when its bench_args are
    this run's arguments (kind "ecc": the summary of `bench.py --registration`, whose launches the configs2 block of the
    default line repeats).  Returns (bytes or None, source or None)."""
    import glob
    prof = os.environ.get("UPSP_BENCH_TRAFFIC_JSON") if kind == "bench" else None
    if not prof:
        cands = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_%s_summary.json" % kind)))
        prof = cands[-1] if cands else None
    if not (prof and os.path.exists(prof)):
        return None, None
    pj = json.load(open(prof))
    mine, skip = [], False
    for x in sys.argv[1:]:       # arguments that do not change the launches: baseline / stress switches, the step counts
        if skip:
            skip = False
        elif x in ("--steps", "--warmup", "--gpus"):
            skip = True
        elif not (x in ("--no-cpu-baseline", "--no-reraycast") or x.startswith(("--steps=", "--warmup=", "--gpus="))):
            mine.append(x)
    want = mine + (["--registration"] if kind == "ecc" and "--registration" not in mine else [])
    if pj.get("bench_args", "").split() == want and world == 1 and kernel in pj.get("traffic_bytes_per_launch", {}):
        return pj["traffic_bytes_per_launch"][kernel], os.path.relpath(os.path.abspath(prof), ROOT)
    return None, None


def merge_ecc_labels(rep):
    """timing_report(spread=True) with one more entry, "ecc_sums_kernel" = the identity and the general sums launches
    together (calls, total ms, min, median of the more frequent kind, max)."""
    parts = [rep[k] for k in ("ecc_sums_identity", "ecc_sums_general") if k in rep]
    if parts:
        most = max(parts, key=lambda v: v[0])
        rep = dict(rep)
        rep["ecc_sums_kernel"] = (sum(v[0] for v in parts), sum(v[1] for v in parts), min(v[2] for v in parts), most[3],
                                  max(v[4] for v in parts))
    return rep


def host_feed_rate(pipe, frames, N, size, pix, chunk=64, nchunks=16):
    """Frames that arrive from HOST memory (the reference's read-ahead thread,
    psp_process.cpp:867-1007): 12-bit packed MRAW bytes in pinned staging slots -> hipMemcpyAsync on the
    feed's copy stream -> upsp_unpack_12bit -> the frame loop, chunk k + 1 uploading while chunk k is
    processed.  PCIe-inclusive rate, reported beside the headline (which has the frames resident)."""
    import torch
    from upsp_processing_amd import engine, video
    npx = size * size
    fb = npx * 3 // 2
    feed = video.FrameFeed(chunk * fb, 3)
    # pack `chunk` frames once on the host (pack_12bpp layout, python/upsp/video/util.py) ...
    px = frames[:chunk].cpu().view(torch.int16).numpy().view(np.uint16).reshape(chunk, -1)
    packed = np.empty((chunk, fb), np.uint8)
    packed[:, 0::3] = px[:, 0::2] >> 4
    packed[:, 1::3] = ((px[:, 0::2] & 0x0F) << 4) | (px[:, 1::2] >> 8)
    packed[:, 2::3] = px[:, 1::2] & 0xFF

    def first_fill(dst):
        dst[:chunk * fb] = packed.reshape(-1)
        return chunk * fb
    # ... and leave the bytes in every pinned slot, as a video reader's read() would (the timed loop
    # re-commits the slots: what is measured is PCIe + unpack + frame loop, not a host memcpy)
    for _ in range(feed.nslots):
        feed.upload(first_fill)
        feed.release()
    F = chunk * nchunks
    rt = torch.empty((N, engine.series_ld(F)), dtype=torch.float32, device="cuda")[:, :F]
    pipe.reset()
    pipe.set_projection(0, pix)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(nchunks):
        d = feed.upload(lambda dst: chunk * fb)
        fr = video.unpack_12bit(d.view(chunk, fb), size, size)
        feed.release()
        pipe.process(fr, first_frame=k * chunk, rows_t=rt, col0=k * chunk, want_rows=False)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    feed.close()
    return {"frames_per_s": F / dt, "pcie_GBps": F * fb / dt / 1e9, "frames": F,
            "how": "12-bit packed frames (1.5 B/pixel) from 3 pinned slots of %d frames: hipMemcpyAsync on a copy "
                   "stream -> unpack in HBM -> frame loop; uploads overlap the processing" % chunk}


def multi_camera_main(a):
    """configs[4] shape on ONE GPU (BASELINE: 4-camera multi-view, 5 M-triangle mesh; all cameras of a frame are kept
    on one GPU): step = a projection build per camera + AverageViews weights + skipped nodes + the weighted frame loop
    over F frame sets (pass A per camera, one whole-row pass B: sol = sum_c w_c * frame_c[pix_c], camera order,
    psp_process.cpp:1771-1843) + finals."""
    import torch
    from upsp_processing_amd import _capi, engine, synthetic as syn
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (no CPU fallback)")
    torch.cuda.set_device(0)
    _capi.lib()
    C, size, F = a.cameras, a.size, a.frames
    if a.small:
        verts, tris = syn.tunnel_model_quad(64, 24)
        F = min(F, 256)
    elif a.model == "5m":
        verts, tris = syn.tunnel_model_quad(576, 205)      # 4 990 104 triangles, 2 495 058 nodes
    else:
        verts, tris = syn.tunnel_model_quad()
    s9, tn = syn.soup(verts, tris)
    nrm = syn.node_normals(verts, tris)
    N = verts.shape[0]
    npx = size * size
    cds = [syn.pinhole_camera(size, size, center=(0, 0, 20), half_extent=6.0, azimuth_deg=360.0 * c / C) for c in range(C)]
    cams = [_capi.make_camera(cd["K"], cd["dist"], cd["R"], cd["t"], size, size) for cd in cds]
    centers = np.array([engine.camera_center(c) for c in cams])
    bvh = engine.BVH(s9)
    d_nodes, d_nrm, d_tn = [torch.as_tensor(x).cuda() for x in (verts, nrm, tn)]
    bvh.set_tri_nodes(d_tn, N)
    frames = [syn.synth_frames_torch(F, size, size, first=100 * c, hot=True) for c in range(C)]
    n_sample = min(F, 32)
    sample = [fr[:n_sample].cpu().view(torch.int16).numpy().view(np.uint16).copy() for fr in frames] if not a.no_cpu_baseline else None
    restorers = [hot_pixel_restorer(fr) for fr in frames]
    pipe = engine.FramePipeline(C, size, size, N)
    pipe.set_row_padding(True)      # (columns F .. pitch of rows_t are padding; the several-camera row pass uses them when it pays)
    rows_t = torch.empty((N, engine.series_ld(F, whole_rows=True)), dtype=torch.float32, device="cuda")[:, :F]
    ev = lambda: torch.cuda.Event(enable_timing=True)
    ev_log, last = [], {}
    torch.cuda.synchronize()
    side = None if a.serial else torch.cuda.Stream(priority=-1)
    # The builds of the cameras do not depend on each other (create_projection_mat per camera, psp_process.cpp:1586-1660): each on
    # a high-priority stream and a BVH handle of its own (engine.BVH.share: same tree, own query scratch) -- four latency-bound
    # chains side by side instead of one after the other (one stream, one handle: 8.03 against 7.26 ms per step, round 5).
    concurrent = side is not None
    bvhs = [bvh] + [bvh.share() for _ in cams[1:]] if concurrent else [bvh] * C
    sides = [side] + [torch.cuda.Stream(priority=-1) for _ in cams[1:]] if concurrent else [side] * C

    def step(record):
        for r in restorers:                 # new frames arrive: the repaired hot pixels are put back
            r()
        e = [ev() for _ in range(3)]
        e[0].record()
        main = torch.cuda.current_stream()
        if side is not None:
            # the builds of a step read the model and the cameras only: on high-priority streams of their own they run beside the
            # PREVIOUS step's frame loop (see main(): no wait in front of them; the consumer side is ordered below)
            per_cam = []
            for c, cam in enumerate(cams):
                with torch.cuda.stream(sides[c]):
                    per_cam.append(engine.build_projection(bvhs[c], cam, d_nodes, d_nrm, d_tn, 70.0, counts=False)["pix"])
            for c in range(1, C):
                if sides[c] is not side:
                    side.wait_stream(sides[c])
                    per_cam[c].record_stream(side)
            with torch.cuda.stream(side):
                pix = torch.stack(per_cam)
                w = engine.projection_weights(pix, d_nodes, d_nrm, centers, "average_view")  # adjust_projection_for_weights
            main.wait_stream(side)
            pix.record_stream(main)
            w.record_stream(main)
        else:
            pix = torch.stack([engine.build_projection(bvh, cam, d_nodes, d_nrm, d_tn, 70.0, counts=False)["pix"] for cam in cams])
            w = engine.projection_weights(pix, d_nodes, d_nrm, centers, "average_view")      # adjust_projection_for_weights
        e[1].record()
        pipe.reset(deferred=True)
        for c in range(C):
            pipe.set_projection(c, pix[c], w[c])
        pipe.process(frames, first_frame=0, rows_t=rows_t, want_rows=False)
        avg, rms = pipe.finalize(F)
        e[2].record()
        if record:
            ev_log.append(e)
            last.update(pix=pix, w=w)
        return avg

    for _ in range(a.warmup):
        step(False)
    torch.cuda.synchronize()
    with quiet_gc():
        t0 = time.perf_counter()
        for _ in range(a.steps):
            step(True)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    for b in set(bvhs):
        b.check()
    _capi.timing_enable(True)
    for _ in range(a.steps):
        step(False)
    torch.cuda.synchronize()
    _capi.timing_enable(False)
    rep = _capi.timing_report(spread=True)
    pix, w = last["pix"], last["w"]
    active = [int(torch.unique(pix[c][pix[c] >= 0]).numel()) for c in range(C)]
    seen = int(((pix >= 0).sum(0) > 0).sum().item())
    seen_by = float((pix >= 0).sum().item()) / max(seen, 1)
    # algorithmic bytes of the pass-B launch: every row written once (4 B x N x F), the per-node scalars of every camera
    # once (index + weight), every active pixel's series read once (2 B x F each)
    row_bytes = 4 * N * F
    alg_b = row_bytes + 8 * N * C + 2 * F * sum(active)
    per_step = {"node_rows_multi_kernel": alg_b, "scan_compact_kernel": C * F * 2 * npx}
    kernels = {}
    for name, (calls, total, lo, med, hi) in rep.items():
        k = {"calls_per_step": calls / a.steps, "ms_per_step": total / a.steps, "avg_launch_ms": total / max(calls, 1)}
        if name in per_step and total:
            k["algorithmic_bytes_per_step"] = per_step[name]
            k["achieved_GBps"] = per_step[name] / (total / a.steps * 1e-3) / 1e9
            k["launch_ms_min_median_max"] = [lo, med, hi]
        kernels[name] = k
    dom = max((n for n in kernels if "achieved_GBps" in kernels[n]), key=lambda n: kernels[n]["ms_per_step"])
    dk = kernels[dom]
    traffic, traffic_src = None, None
    prof = os.environ.get("UPSP_BENCH_TRAFFIC_JSON")
    if prof and os.path.exists(prof):
        pj = json.load(open(prof))
        if dom in pj.get("traffic_bytes_per_launch", {}):
            traffic, traffic_src = pj["traffic_bytes_per_launch"][dom], os.path.relpath(os.path.abspath(prof), ROOT)
    ms_step = dt / a.steps * 1e3
    out = {
        "metric": "frames/s", "value": F * a.steps / dt, "unit": "frames/s", "n_gpus": 1, "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": ms_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "camera_frames_per_s": C * F * a.steps / dt,
        "config": {"workload": "configs[4] shape on one GPU: %d cameras x %d frames x %dx%d u16, %d-tri tunnel model (%d nodes), "
                               "raycast+weighted projection; a frame = one frame of every camera" % (C, F, size, size, tris.shape[0], N),
                   "cameras": C, "frames_per_camera": F, "nodes": N, "triangles": int(tris.shape[0]),
                   "nodes_seen": seen, "cameras_per_seen_node": seen_by, "active_pixels_per_camera": active,
                   "parallelism": "one GPU", "schedule": ("per camera: projection build; pass A per camera; one whole-row pass B over all cameras" if a.serial else
                                ("the projection builds of a step on high-priority streams of their own, %s (beside the previous step's frame "
                                 "loop); pass A per camera; one whole-row pass B over all cameras") % ("the cameras side by side" if concurrent else "camera after camera"))},
        "breakdown_ms": {"projection_builds_and_weights": float(np.mean([e[0].elapsed_time(e[1]) for e in ev_log])),
                         "frame_loop_and_finals": float(np.mean([e[1].elapsed_time(e[2]) for e in ev_log]))},
        "roofline": {"kernel": dom, "bound": "hbm", "achieved": dk["achieved_GBps"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": dk["achieved_GBps"] / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                     "algorithmic_bytes_per_launch": dk["algorithmic_bytes_per_step"] / max(dk["calls_per_step"], 1),
                     "avg_launch_ms": dk["avg_launch_ms"], "launches_per_step": dk["calls_per_step"]},
        "pass_b_row_GBps": row_bytes / (kernels["node_rows_multi_kernel"]["ms_per_step"] * 1e-3) / 1e9 if "node_rows_multi_kernel" in kernels else None,
        "kernels": kernels,
    }
    roofline_extras(out["roofline"], dict(per_step), ms_step)
    if not a.no_cpu_baseline:
        # oracle (CPU port): projection of every camera on the full model, weights, and the weighted loop on a bounded
        # sample of frame sets (one frame set per thread, like the reference's OpenMP loop)
        from concurrent.futures import ThreadPoolExecutor
        from oracle import oracle as orc
        cores = usable_cpus()
        obv = orc.OracleBVH(s9)
        t0 = time.perf_counter()
        opix = np.stack([orc.create_projection(obv, orc.make_camera(cd["K"], cd["dist"], cd["R"], cd["t"], size, size), verts, nrm, tn,
                                               engine.oblique_threshold(70.0), threads=cores)["pix"] for cd in cds]).astype(np.int32)
        t_proj = time.perf_counter() - t0
        ow = orc.adjust_weights(opix, np.ones((C, N), np.float32), verts, nrm, centers, 1)      # 1 = AverageViews
        osk = orc.skipped_nodes(opix)

        def one(f):
            sol = None
            for c in range(C):
                img, _ = orc.fix_hot_pixels(sample[c][f])
                cs = orc.project_frame(img, opix[c], ow[c])
                sol = cs if sol is None else (sol + cs).astype(np.float32)
            sol[osk] = np.nan
            return sol
        t0 = time.perf_counter()
        with ThreadPoolExecutor(cores) as ex:
            rows_o = list(ex.map(one, range(n_sample)))
        t_loop = time.perf_counter() - t0
        out["cpu_baseline"] = {"value": F / (t_proj + t_loop / n_sample * F), "unit": "frames/s", "cores": cores, "kind": "port",
                               "sample": "oracle/ (C, %d threads): projection of the %d cameras on the full model (%.1f s) + the weighted loop on %d "
                                         "frame sets (%.2f s); extrapolated to %d frame sets" % (cores, C, t_proj, n_sample, t_loop, F)}
        gw = w.cpu().numpy()
        p2 = engine.FramePipeline(C, size, size, N, fused_scan=1)
        for c in range(C):
            p2.set_projection(c, pix[c], w[c])
        rt = torch.empty((N, engine.series_ld(n_sample, whole_rows=True)), dtype=torch.float32, device="cuda")[:, :n_sample]
        p2.process([torch.as_tensor(sm.view(np.int16)).view(torch.uint16).cuda() for sm in sample], 0, rows_t=rt, want_rows=False)
        gr = rt.cpu().numpy()
        gs, gss = [x.cpu().numpy() for x in p2.accumulators()]
        ro = np.stack(rows_o)
        ok = ~np.isnan(ro[0])
        so, sso = ro[:, ok].astype(np.float64).sum(0), (ro[:, ok] * ro[:, ok]).astype(np.float64).sum(0)
        seen2 = (opix >= 0).sum(0) >= 2
        checks = {
            "projection_pix_all_cameras": bool(np.array_equal(pix.cpu().numpy(), opix)),
            # AverageViews weights: f64 acos -> f32, a few 1e-7 relative (tests/test_projection_gpu.py)
            "weights_rel_4e-7": bool(np.abs(gw - ow).max() <= 4e-7),
            "series_rows_%d_frame_sets_for_gpu_weights" % n_sample: None,
            "nan_rows": bool(np.array_equal(np.isnan(gr[:, 0]), ~ok)),
            "accumulators_rel_1e-12": bool(np.allclose(gs[ok], so, rtol=1e-9) and np.allclose(gss[ok], sso, rtol=1e-9)),
        }
        # the series against the oracle's weighted sum evaluated with the GPU's weights (the weights themselves are compared
        # above, to 1 ulp): bit for bit
        def one_g(f):
            sol = None
            for c in range(C):
                img, _ = orc.fix_hot_pixels(sample[c][f])
                cs = orc.project_frame(img, opix[c], gw[c])
                sol = cs if sol is None else (sol + cs).astype(np.float32)
            return sol
        with ThreadPoolExecutor(cores) as ex:
            rg = np.stack(list(ex.map(one_g, range(n_sample))))
        checks["series_rows_%d_frame_sets_for_gpu_weights" % n_sample] = bool(np.array_equal(gr[ok].T.view(np.int32), rg[:, ok].view(np.int32)))
        acc_ok = np.allclose(gs[ok], rg[:, ok].astype(np.float64).sum(0), rtol=1e-12) and np.allclose(
            gss[ok], (rg[:, ok] * rg[:, ok]).astype(np.float64).sum(0), rtol=1e-12)
        checks["accumulators_rel_1e-12"] = bool(acc_ok)
        out["parity_checked"] = all(checks.values())
        out["parity"] = checks
        out["config"]["nodes_seen_by_two_or_more"] = int(seen2.sum())
        if not out["parity_checked"]:
            print(json.dumps(out), flush=True)
            raise SystemExit("bench.py: GPU results differ from the oracle: %r" % (checks,))
    else:
        out["cpu_baseline"] = None
    print(json.dumps(out), flush=True)


def main():
    a = parse()
    if a.cameras > 1:
        if a.gpus != 1:
            raise SystemExit("bench.py --cameras: one GPU (all cameras of a frame are kept on one GPU)")
        return multi_camera_main(a)
    if a.gpus > 1 and "RANK" not in os.environ and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(a.gpus))
    run(a)


def configs3_block(a, world, N, npix):
    """BASELINE configs[3] at its OWN size inside the N > 1 line: 100 000 frames sharded over the ranks (apportion,
    cpp/exec/psp_process.cpp:611-624, :1519-1529), every rank's share resident in HBM, ONE step through the chunked pixel-series
    exchange with real peers (the --config3-share code path: pass A per <= 1024-frame chunk, a block per peer and chunk, the owner's
    pass B over all frames of the run, the all-reduce of the sums; global_transpose :707-771).  The frames a rank can hold are
    bounded by its HBM: 2 MiB of frame + 4 B x N of its series slice per frame of the rank, + the exchange's buffers -- when 70 %
    of the free memory does not hold the share (two ranks: 50 000 frames = 230 GB), the block runs on what fits and says so.
    UPSP_BENCH_CONFIGS3_FRAMES: another total (the tests)."""
    import copy
    import torch
    import torch.distributed as dist
    total = int(os.environ.get("UPSP_BENCH_CONFIGS3_FRAMES", "100000"))
    share = total // world
    per_frame = npix * 2 + N * 4 + (1 << 19)
    free_b = torch.cuda.mem_get_info()[0]
    fit = int(0.7 * free_b / per_frame) // 64 * 64
    t = torch.tensor([min(share, fit)], dtype=torch.int64, device="cuda")
    dist.all_reduce(t, op=dist.ReduceOp.MIN)
    F3 = int(t.item())
    if F3 < 64:
        return {"skipped": "%.0f GB of HBM free on the fullest rank: no room for a share of configs[3]" % (free_b / 1e9)}
    a3 = copy.copy(a)
    a3.config3_share, a3.frames, a3.steps, a3.warmup = True, F3, 1, 1
    a3.no_cpu_baseline = a3.no_reraycast = True
    blk = run(a3, nested=True)
    keep = ("value", "unit", "ms_per_step", "steps", "warmup", "ms_per_step_rank_min_max", "rccl_nranks", "rccl_bound", "rccl_library",
            "exchange_self_check", "exchange_finals_check", "exchange_bytes_per_step")
    out = {k: blk[k] for k in keep if k in blk}
    out["workload"] = "configs[3]: %d frames x %dx%d u16 sharded over %d ranks (%d per rank), 1 M-tri model, time-series exchange at the end of the step" % (
        F3 * world, a.size, a.size, world, F3)
    out["frames_per_rank"] = F3
    out["exchange"] = blk["config"].get("exchange")
    if F3 < share:
        out["note"] = "configs[3]'s share is %d frames per rank; %.0f GB of HBM free hold %d" % (share, free_b / 1e9, F3)
    return out


def run(a, nested=False):
    if a.config3_share:
        a.force_chunked = True
        if a.frames == 1000:
            a.frames = 12500
        a.no_reraycast = True
    import torch
    import torch.distributed as dist
    from upsp_processing_amd import _capi, engine, synthetic as syn, distributed as D

    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != a.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE is %d" % (a.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (no CPU fallback)")
    # rehearsal of the N > 1 code path on a one-GPU box (tests): UPSP_BENCH_BACKEND=gloo for torch.distributed's rendezvous with every
    # rank on cuda:0 (UPSP_BENCH_ONE_GPU=1) and UPSP_RCCL_LIBRARY naming an RCCL that accepts that (tests/shim).  The driver's runs
    # use RCCL, one rank per GPU.
    backend = os.environ.get("UPSP_BENCH_BACKEND", "nccl")
    if os.environ.get("UPSP_BENCH_ONE_GPU"):
        local = 0
    torch.cuda.set_device(local)
    force_coll = world == 1 and bool(os.environ.get("UPSP_FORCE_COLLECTIVES"))   # one-rank group: RCCL calls on one GPU
    if force_coll:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        with socket.socket() as sck:
            sck.bind(("127.0.0.1", 0))
            os.environ.setdefault("MASTER_PORT", str(sck.getsockname()[1]))
    if (world > 1 or force_coll) and not nested:
        import datetime
        try:
            # (a rank that never shows up must end the job with a message, not hang it: 5-minute rendezvous limit)
            tmo = datetime.timedelta(seconds=int(os.environ.get("UPSP_BENCH_INIT_TIMEOUT", "300")))
            if backend == "nccl":
                dist.init_process_group("nccl", rank=rank, world_size=world, timeout=tmo,
                                        device_id=torch.device("cuda", local))
            else:
                dist.init_process_group(backend, rank=rank, world_size=world, timeout=tmo)
            # first collective = the communicator really exists on every rank
            probe = torch.ones(1, device="cuda")
            dist.all_reduce(probe)
            torch.cuda.synchronize()
            if int(probe.item()) != world:
                raise RuntimeError("all_reduce over %d ranks returned %g" % (world, float(probe.item())))
            # the library's own communicator (upsp_comm_create from an id broadcast over the group): the exchanges of this run have
            # no other path -- if it does not come up on every rank the run ends here
            D.lib_comm()
        except Exception as e:                                   # noqa: BLE001
            print("bench.py: rank %d of %d could not join the %s process group: %s: %s" % (rank, world, backend, type(e).__name__, e),
                  file=sys.stderr, flush=True)
            raise SystemExit(3)
    _capi.lib()

    size = a.size
    F = a.frames
    if a.small:
        verts, tris = syn.tunnel_model_quad(64, 24)
        if not nested:
            F = min(F, 64)
    elif a.fill_frame:
        verts, tris = syn.cube_sphere(289, 6.0)   # 1 002 252 triangles, 501 128 nodes: a sphere that fills the frame
    elif a.model == "uv":
        verts, tris = syn.tunnel_model()          # 1 001 520 triangles, 500 766 nodes, polar fans
    else:
        verts, tris = syn.tunnel_model_quad()     # 1 001 904 triangles, 500 958 nodes, valence <= 6
    s9, tn = syn.soup(verts, tris)
    nrm = syn.node_normals(verts, tris)
    N = verts.shape[0]
    cd = syn.pinhole_camera(size, size, center=(0, 0, 20), half_extent=6.0, fill=0.95 if a.fill_frame else 0.7)
    cam = _capi.make_camera(cd["K"], cd["dist"], cd["R"], cd["t"], size, size)

    bvh = engine.BVH(s9)
    d_nodes = torch.as_tensor(verts).cuda()
    d_nrm = torch.as_tensor(nrm).cuda()
    d_tn = torch.as_tensor(tn).cuda()
    bvh.set_tri_nodes(d_tn, N)      # createBVH(model, triNodes): once per model, like the BVH itself
    # static content of the frames: model silhouette / background, fiducial discs
    proj0 = engine.build_projection(bvh, cam, d_nodes, d_nrm, d_tn, 70.0, counts=False)
    pix0 = proj0["pix"].cpu().numpy()
    layout = None if a.plain_frames else syn.scene_layout(pix0, size, size)
    n_active = int(np.unique(pix0[pix0 >= 0]).size)
    # this rank's frames (global frame index = rank*F + f), resident in HBM
    frames = torch.empty((F, size, size), dtype=torch.uint16, device="cuda")
    chunk = 50
    for f0 in range(0, F, chunk):
        syn.synth_frames_torch(min(chunk, F - f0), size, size, first=rank * F + f0, out=frames[f0:f0 + chunk],
                               layout=layout, hot=not a.plain_frames)
    n_sample = min(F, 256)
    sample = (frames[:n_sample].cpu().view(torch.int16).numpy().view(np.uint16).copy()
              if (rank == 0 and world == 1 and not a.no_cpu_baseline) else None)   # before any in-place repair
    restore_hot = hot_pixel_restorer(frames)        # the hot pixels are put back before every step (inside the timed region)
    shard = D.Shard(F * world, N, rank, world)
    pipe = engine.FramePipeline(1, size, size, N, registration=int(a.registration),
                                fused_scan=2 if a.two_kernel else 0)
    if a.registration:
        pipe.set_reference(0, frames[0].to(torch.float32))   # raw first frame as ECC template
    # node-major time series [N, F]; one process() call writes every row piece whole
    streamed = not (a.two_kernel or a.registration)
    # (the streamed schedule -- and registration as the last image stage -- write every row piece whole, one pass B per <= 1024 frames)
    ld = engine.series_ld(F, whole_rows=streamed or (a.registration and not a.two_kernel))
    chunked = world > 1 or a.force_chunked          # --force-chunked: exercise the N>1 loop on one GPU
    rows_t = torch.empty((N, ld), dtype=torch.float32, device="cuda")[:, :F] if not chunked else None
    # (columns F .. ld of that allocation are padding: the row pass may end every row on a whole 128-byte line)
    row_padding = rows_t is not None
    pipe.set_row_padding(row_padding)
    # pass A in two launches where it runs beside the projection build (the default schedule): the tiles nobody reads as one-wave
    # workgroups without LDS -- step 0.799 -> 0.773 ms; alone on the device (--serial) the one-launch form is the faster one
    # (the N > 1 loop keeps one launch: 1.07 against 1.10 ms per step in the one-rank rehearsal -- its pass A runs beside less of the build)
    scan_split_on = not a.serial and not a.registration and not chunked
    if scan_split_on:
        pipe.set_scan_split(True)
    torch.cuda.synchronize()

    ev = lambda: torch.cuda.Event(enable_timing=True)
    t_ray, t_frames, t_xchg, host_log = [], [], [], []
    last_pix = [None]

    # N > 1: the frame loop runs in K chunks and the all-to-all of chunk k is issued
    # asynchronously while chunk k+1 is being processed (distributed.TimeSeriesExchange)
    # one camera, no weights, no filter: the series values are exact 16-bit integers, so the
    # travelling rows are produced and sent as u16 (half the bytes) and widened by the receiver
    u16_wire = chunked and not a.f32_wire
    pixel_wire = chunked and not a.row_wire and not a.f32_wire
    # Pixel-series wire: TWO exchanges used in turn.  A rank's run is many steps long (configs[3]: 12 500 frames per rank), and
    # what a step leaves behind -- its last chunk still on the links, the owner's pass B -- does not have to be waited for
    # before the next step's frames are scanned: step k's exchange is finished after step k + 1's chunks have been submitted.
    # (between GPUs only, or on request: on ONE GPU nothing waits for a link, and the later pass B finds its compact series
    #  pushed out of the Infinity Cache by the next step's frames -- 1.56 against 1.45 ms per step, measured)
    deferred = chunked and pixel_wire and not a.sync_exchange and (world > 1 or a.defer_exchange)
    # Pass A runs ONCE for all frames of the rank, on the candidate-pixel map, beside the projection build -- the arrangement of
    # the one-GPU loop -- and one block per peer goes out; the owner's pass B reads the blocks where they arrive.  (--chunk-scan:
    # pass A per chunk after the build, every chunk's sends beside the next chunk's scan -- the schedule of rounds 3 / early 4;
    # one GPU through RCCL, same call: 1.61-1.63 / 1.74 ms per step finished in the step / deferred, against 1.46-1.48 / 1.46.)
    px_once = (chunked and pixel_wire and not a.chunk_scan and not a.config3_share and F <= min(1024, pipe.series_frames_max()))
    K = max(1, a.chunks if a.chunks > 0 else (1 if px_once else 4)) if chunked else 1
    if chunked and (a.config3_share or F > 1024 * K):
        # every chunk within one pass A group (<= 1024 frames; the cuts sit on 64-frame boundaries)
        K = D.chunk_count(shard.frame_count, 1024)
    exch = D.TimeSeriesExchange(shard, K, wire12=a.wire12) if chunked else None
    chunk_bufs = ([torch.empty((N, exch.my_chunk(k)[1]), dtype=torch.uint16 if u16_wire else torch.float32,
                               device="cuda") for k in range(K)] if chunked else None)
    ev_log = []
    first_step = [True]
    exchs = [exch, D.TimeSeriesExchange(shard, K, wire12=a.wire12)] if deferred else [exch]
    ex_state = {"step": 0, "pending": None, "first": [True, True], "last_done": exch}

    def drain():
        """finish the exchange the previous step left in flight (pass B of its frames: series + its slice of the sums)"""
        if ex_state["pending"] is not None:
            s_, ss_ = pipe.accumulators()
            ex_state["pending"].finish_pixels(s_, ss_)
            ex_state["last_done"] = ex_state["pending"]
            ex_state["pending"] = None

    # Pass A reads the frames and needs only the CANDIDATE pixels (known after step 1 of create_projection_mat), the ray casting
    # is a latency-bound chain of dependent fetches: side by side they take 0.59-0.63 ms where one after the other they take
    # 0.29 + 0.35-0.39 (LAB_NOTES.md, three alternations in one call: step 1.10-1.13 against 1.16-1.18 ms)
    overlap = not a.serial and not a.registration and streamed and F <= 1024 and (not chunked or px_once)
    # THE DEFAULT STEP IS ONE LIBRARY CALL: upsp_pipeline_step (engine.FramePipeline.step).  The build of the step -- candidate
    # pixels, active-pixel map, ray casting, hand-over, node -> row sweep, the finals of the step before -- runs on a high-priority
    # stream the pipeline owns, pass A + repair + pass B on this process's stream, every ordering event inside the library
    # (include/upsp_gpu.h; LAB_NOTES.md sections 7-12 hold the measurements behind the arrangement: build on the side stream at
    # high priority 1.107 against 1.151-1.208 ms for the alternatives in round 3, the map in front of the build 0.90-0.96 -> 0.86,
    # three launches on the main stream 0.877 -> 0.839, pass A in two launches 0.799 -> 0.773).  What this file still does per step:
    # put the hot pixels back (frames hook: the bench repairs the same resident frames every step) and, at N > 1, its exchange.
    lean = overlap
    # configs[2]: the build of a step on a stream of its own -- it runs beside the previous step's registration.  (The side stream
    # must be a HIGH-priority one to get a hardware queue of its own: a normal-priority torch stream shares the default stream's
    # queue on this runtime.)
    reg_side = a.registration and not a.serial and not chunked
    side = torch.cuda.Stream(priority=-1) if reg_side else None
    # (N > 1, pixel-series wire: the same call without pass B -- the owner of a node runs it; the all-reduce + finals of a step and the
    #  exchange's pixel table ride on the side stream through the tail hook)
    lean_px = lean and chunked and pixel_wire and px_once
    lean_st = {"finals_due": False}
    finals_buf = (torch.empty(N, dtype=torch.float32, device="cuda"), torch.empty(N, dtype=torch.float32, device="cuda")) if lean else None

    def lean_finals():
        """N > 1: finals of the step before (the accumulators as its pass B left them) -- the all-reduce of the two sum vectors in
        front of them, on the stream this is called on: the side stream, not the frame loop's (only the finals need the complete
        sums, and a collective in the main stream is a rendezvous of all ranks in front of every pass A)."""
        if lean_st["finals_due"]:
            D.allreduce_sums(*pipe.accumulators())
            lean_st["avg"] = pipe.finalize(F * world)[0]
            lean_st["finals_due"] = False

    def step_lean(record, events):
        e = [ev() for _ in range(4)] if events else None
        if events:
            e[0].record()
            e[1].record()
        ht = [time.perf_counter()] * 2

        def tail(_st):
            # behind the node -> row sweep of the new projection and the end of the previous step, on the pipeline's side stream
            lean_finals()
            if lean_px:
                # which pixel rows go where: from the node -> row table and the skipped flags the sweep just wrote
                # (after the first step only compared with the exchange's own copy, on this stream)
                wh = ex_state["step"] % len(exchs)
                tabs = pipe.row_tables()
                exchs[wh].set_pixels(tabs["node_k"], tabs["skipped"], assume_same=not ex_state["first"][wh])
                ex_state["first"][wh] = False

        pipe.step(bvh, cam, d_nodes, d_nrm, d_tn, frames, rows_t=None if chunked else rows_t, first_frame=rank * F,
                  finals=None if chunked else finals_buf, nframes_total=F * world, frames_hook=lambda _st: restore_hot(),
                  tail_hook=tail if chunked else None)
        if chunked:
            pipe.reset(deferred=True)               # the sums of whichever exchange is finished next start from zero
            which = ex_state["step"] % len(exchs)
            ex = exchs[which]
            ex_state["step"] += 1
            ex.k = 0
            tab = pipe.pixel_series(frames)         # pass A ran beside the build: the rows of the nodes in that buffer
            for k in range(K):
                ex.submit_pixels(tab, col0=ex.my_chunk(k)[0])
            if deferred:
                drain()                             # the PREVIOUS step's series and sums, now that this step's block is on its way
                ex_state["pending"] = ex
            else:
                ex.finish_pixels(*pipe.accumulators())
            lean_st["finals_due"] = True            # (all-reduce + finals: on the side stream of the next step, or behind the loop)
            pipe.step_mark_end()
        ht += [time.perf_counter()] * 3
        if events:
            e[2].record()
            e[3].record()
        if record and events:
            ev_log.append(e)
            host_log.append([(b - a) * 1e3 for a, b in zip(ht[:-1], ht[1:])])
        if record:
            last_pix[0] = pipe.current_projection(0)
        return None

    def step(record, events=True):
        if lean:
            return step_lean(record, events)
        e = [ev() for _ in range(4)]
        restore_hot()
        e[0].record()
        ht = [time.perf_counter()]
        main = torch.cuda.current_stream()
        if reg_side:
            with torch.cuda.stream(side):           # beside the previous step's registration
                proj = engine.build_projection(bvh, cam, d_nodes, d_nrm, d_tn, 70.0, counts=False)
        else:
            proj = engine.build_projection(bvh, cam, d_nodes, d_nrm, d_tn, 70.0, counts=False)   # no host sync
        e[1].record()
        ht.append(time.perf_counter())
        pipe.reset(deferred=True)
        if reg_side:
            main.wait_stream(side)
            for t in proj.values():
                if isinstance(t, torch.Tensor) and t.is_cuda:
                    t.record_stream(main)
        pipe.set_projection(0, proj["pix"])
        if not chunked:
            pipe.process(frames, first_frame=rank * F, rows_t=rows_t, want_rows=False)
        elif pixel_wire:
            # the sender runs pass A (+ hot-pixel repair) per chunk and ships the active pixels' series; the owner of a node
            # runs pass B over all frames (exch.finish_pixels below)
            which = ex_state["step"] % len(exchs)
            ex = exchs[which]
            ex_state["step"] += 1
            ex.k = 0
            tab = pipe.pixel_series(None)       # node -> compact row of this projection
            ex.set_pixels(tab["node_k"], engine.skipped_nodes(proj["pix"], want_count=False, as_bool=False)[0], assume_same=not ex_state["first"][which])
            ex_state["first"][which] = False
            for k in range(K):
                c0, fc = ex.my_chunk(k)
                ex.submit_pixels(pipe.pixel_series(frames[c0:c0 + fc]) if fc else tab)
            if deferred:
                drain()                             # the PREVIOUS step's series and sums, now that this step's chunks are on their way
                ex_state["pending"] = ex
        else:
            exch.k = 0
            # rows of nodes no camera sees are NaN on every rank: they do not travel, and pass B writes
            # the travelling rows packed (row map) straight into the send buffers.  The projection is
            # rebuilt every step and is the same every step: the travelling set is derived (one host
            # read of W counters) in the first step and only verified on the device afterwards.
            exch.set_skipped(engine.skipped_nodes(proj["pix"], want_count=False, as_bool=False)[0], assume_same=not first_step[0])
            first_step[0] = False
            pipe.set_row_map(exch.row_map())
            nrows = exch.packed_rows()
            for k in range(K):
                c0, fc = exch.my_chunk(k)
                buf = chunk_bufs[k][:nrows]
                if fc:
                    pipe.process(frames[c0:c0 + fc], first_frame=rank * F + c0, rows_t=buf, want_rows=False)
                exch.submit(buf, packed=True)
        e[2].record()
        ht.append(time.perf_counter())
        s, ss = pipe.accumulators()
        if pixel_wire and not deferred:
            exchs[0].finish_pixels(s, ss)         # pass B of this rank's nodes over all frames: series + its slice of the sums
        D.allreduce_sums(s, ss)
        if chunked and not pixel_wire:
            exch.finish()
        avg, rms = pipe.finalize(F * world)
        e[3].record()
        ht.append(time.perf_counter())
        if record:           # events are read after the timed loop: no host sync inside it
            ev_log.append(e)
            host_log.append([(b - a) * 1e3 for a, b in zip(ht[:-1], ht[1:])])
            last_pix[0] = proj["pix"]
        return avg

    def finish_run():
        """what the last step left to do: the deferred exchange's series and sums, and the finals that ride on the NEXT step's side
        stream in the one-call arrangement"""
        if not lean:
            drain()
            return
        pipe.step_finish()                      # (N = 1: the last step's finals; the main stream behind the side stream)
        if chunked:
            lean_finals()                       # finals of the sums the last step's pass B (or its drain) left
            if ex_state["pending"] is not None: # deferred: the last step's own exchange -- its sums start from zero like every step's
                pipe.reset(deferred=True)
                drain()
                lean_st["finals_due"] = True
                lean_finals()

    for _ in range(a.warmup):
        step(False)

    def barrier():
        if world > 1 or force_coll:
            dist.barrier()
        torch.cuda.synchronize()

    barrier()
    with quiet_gc():
        t0 = time.perf_counter()
        for _ in range(a.steps):
            step(True, events=not lean)
        finish_run()                    # (deferred exchange: the last step's series and sums; the last finals: inside the timed region)
        barrier()
        dt = time.perf_counter() - t0
    if lean:
        last_pix[0] = last_pix[0].clone()       # (a view of the pipeline's buffer: the repetitions below build into it again)
    xcheck = fcheck = None
    if chunked:
        for x in exchs:
            x.verify()                  # the travelling set did not change between the steps
        # What came out of the exchange against this rank's own frames, on every rank: for a sample of the nodes this rank
        # owns, the columns of ITS frames in the series slice must be frame[pix] (repaired frames; exact in every wire format)
        # and the rows of nodes no ray sees NaN.  (The oracle comparison of rank 0 in a one-rank run checks values end to end;
        # this one runs at any N and catches a block that landed in the wrong place.)
        o = (ex_state["last_done"] if pixel_wire else exch).out
        n0, nn = shard.my_nodes
        ok = True
        if nn > 0 and F > 0:
            g = torch.Generator(device="cuda")
            g.manual_seed(1234 + rank)
            pick = torch.randperm(nn, generator=g, device="cuda")[:2048]
            pp = last_pix[0][n0:n0 + nn].long()[pick]
            vis = pp >= 0
            want = frames.view(torch.int16).reshape(F, -1)[:, pp.clamp(min=0)].T.to(torch.float32)
            got = o[pick][:, rank * F:(rank + 1) * F]
            ok = bool(torch.equal(got[vis], want[vis])) and bool(torch.isnan(got[~vis]).all())
        fin_ok = True
        if lean and lean_st.get("avg") is not None and nn > 0 and F > 0:
            # the finals of the last exchange (sums all-reduced over the ranks) against the series this rank holds for its own nodes:
            # avg[n] = mean of row n over ALL frames of the run, NaN for the nodes no ray sees
            torch.cuda.synchronize()
            rows = o[pick].double()
            want_avg = rows.mean(1).float()
            got_avg = lean_st["avg"][n0:n0 + nn][pick]
            seen = ~torch.isnan(want_avg)
            fin_ok = bool(torch.isnan(got_avg[~seen]).all()) and bool(torch.allclose(got_avg[seen], want_avg[seen], rtol=1e-6, atol=0.0))
        t = torch.tensor([1 if ok else 0, 1 if fin_ok else 0], device="cuda", dtype=torch.int32)
        if world > 1 or force_coll:
            dist.all_reduce(t, op=dist.ReduceOp.MIN)
        xcheck = bool(t[0].item())
        fcheck = bool(t[1].item()) if lean else None
    bvh.check()                         # no walk ran past its round cap (UPSP_ERR_INTERNAL otherwise)
    pc = engine.projection_counts(bvh)
    primary_rays, retry_nodes = pc["primary_rays"], pc["retry_nodes"]
    # the rays the REFERENCE casts for this camera: one build (not timed) that casts them all, in the reference's
    # order (oblique test last); its entries must be the ones of the timed builds
    pr_ref = engine.build_projection(bvh, cam, d_nodes, d_nrm, d_tn, 70.0, counts=True)
    nrays = pr_ref["nrays"]
    same_entries = bool(torch.equal(pr_ref["pix"], last_pix[0]))
    # the same K steps once more with the library's per-kernel HIP-event timers on
    # (two extra events per launch on the launch stream; kept out of the headline time)
    _capi.timing_enable(True)
    keep_pix = last_pix[0]
    for _ in range(a.steps):
        step(lean)                      # (lean: the per-phase events of the breakdown are recorded here, not in the timed steps)
    last_pix[0] = keep_pix
    finish_run()
    barrier()
    _capi.timing_enable(False)
    for e in ev_log:
        t_ray.append(e[0].elapsed_time(e[1]))
        t_frames.append(e[1].elapsed_time(e[2]))
        t_xchg.append(e[2].elapsed_time(e[3]))
    dt_rank_min = dt_rank_max = dt
    if world > 1:
        tt = torch.tensor([dt, -dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt_rank_max, dt_rank_min = float(tt[0].item()), -float(tt[1].item())
        dt = dt_rank_max

    ms_step = dt / a.steps * 1e3
    total_frames = F * world
    fps = total_frames * a.steps / dt
    ray_ms = float(np.mean(t_ray))
    frm_ms = float(np.mean(t_frames))
    ray_alone_ms = ray_ms
    if overlap:
        # pass A ran beside the build in the timed steps: the rates derived from "the build" and "the frame loop" below take the
        # build's own duration from a few builds timed alone (not part of `value`)
        torch.cuda.synchronize()
        b0, b1 = ev(), ev()
        b0.record()
        for _ in range(5):
            engine.build_projection(bvh, cam, d_nodes, d_nrm, d_tn, 70.0, counts=False)
        b1.record()
        torch.cuda.synchronize()
        ray_alone_ms = b0.elapsed_time(b1) / 5
        frm_ms_share = max(ms_step - ray_alone_ms, 1e-6)
    mrays = nrays / (ray_alone_ms * 1e-3) / 1e6

    # per-kernel durations: HIP events recorded by the library on the launch stream
    # during the timed steps (upsp_timing_enable / upsp_timing_report)
    timing_full = merge_ecc_labels(_capi.timing_report(spread=True))
    timing = {k: v[:2] for k, v in timing_full.items()}
    kernels = {}
    n_retry_rays = 6 * retry_nodes
    scene_bytes = bvh.info["device_bytes"]
    my_chunks = [exch.my_chunk(k)[1] for k in range(K)] if chunked else [F]
    gather_launches = sum(-(-c // 64) for c in my_chunks)
    row_launches = sum(-(-c // 1024) for c in my_chunks)
    series_rows = (exch.pixel_rows()[0] if pixel_wire else exch.packed_rows()) if chunked else N
    series_esz = 2 if u16_wire else 4
    npx = size * size
    per_step_bytes = {
        # 40 B per ray (24 B ray + 16 B hit record) + the scene once per launch
        "projection_kernel<primary>": primary_rays * 40 + scene_bytes,
        # the occluder-witness pass sees every retry ray (ray + verdict); the traversal that follows
        # only the few it leaves undecided (count known to the device only) plus the scene
        "witness_kernels": n_retry_rays * 40,
        "projection_kernel<retry>": scene_bytes,
        # frame unit = 2 MiB frame + 12 B x N (pix 4, weight 4, out 4).  The 2 MiB
        # compulsory full read of the frame belongs to the scan (pass A / hot_scan_kernel); the series
        # writers keep the per-node index in registers across a launch, so per LAUNCH they need
        # 4 B x N x frames written + 8 B x N read once -- counting 12 B x N per FRAME would credit
        # bytes the kernel never has to move.
        # (N > 1: only the rows that travel are stored, as u16 when the values are 16-bit integers)
        "gather_tile_kernel": F * series_esz * series_rows + gather_launches * 8 * N,
        "hot_scan_kernel": F * 2 * npx,
        "scan_compact_kernel": F * 2 * npx,
        "node_rows_kernel": F * series_esz * series_rows + row_launches * 8 * N,
        # SURVEY.md 8(d) with registration: 8 B per pixel and ECC iteration (blurred frame + template,
        # gradients recomputed on the fly), warp 2 B in + 2 B out per pixel, pre-blur 2 B in + 4 B out
        # the warp produces only the pixels a node reads (list): 4 B list entry + 4 x 2 B source + 2 B out each
        "warp_u16_kernel": F * n_active * 14,
        "gauss_pass_kernels": F * 6 * npx,
    }
    # traversal passes: SURVEY.md 8(d)'s 40 B per ray; the scene is NOT credited per launch (a pass of a few thousand
    # rays touches a fraction of it, and it stays cache-resident between the passes)
    per_step_bytes["projection_kernel<primary>"] = primary_rays * 40
    del per_step_bytes["projection_kernel<retry>"]
    if pixel_wire:       # pass B runs on the owner of the nodes, over all frames of the run, f32 rows
        nn_me, f_all = shard.my_nodes[1], F * world
        per_step_bytes["node_rows_kernel"] = f_all * 4 * nn_me + (-(-f_all // 1024)) * 8 * nn_me
    if a.registration:
        st = pipe.ecc_stats()            # average ECC iterations per frame over every step run so far
        ecc_iters_per_frame = st["frame_iterations"] / max(st["frames"], 1)
        per_step_bytes["ecc_sums_kernel"] = ecc_iters_per_frame * F * 8 * npx
        # (the two kernels behind that label: a frame's first iteration starts from the identity warp, the others are general)
        per_step_bytes["ecc_sums_identity"] = F * 8 * npx
        per_step_bytes["ecc_sums_general"] = max(ecc_iters_per_frame - 1.0, 0.0) * F * 8 * npx
        if "ecc_blur_ident_kernel" in timing and "ecc_sums_identity" not in timing:
            # the pre-blur and the identity iteration in one pass: 2 B in + 4 B out + 4 B template per pixel; the sums launches
            # that remain are the general iterations
            per_step_bytes["ecc_blur_ident_kernel"] = F * 10 * npx
            per_step_bytes["ecc_sums_kernel"] = per_step_bytes["ecc_sums_general"]
    for name, (calls, total_ms) in timing.items():
        ms_step_k = total_ms / a.steps
        k = {"calls_per_step": calls / a.steps, "ms_per_step": ms_step_k,
             "avg_launch_ms": total_ms / max(calls, 1)}
        if name in per_step_bytes:
            k["algorithmic_bytes_per_step"] = per_step_bytes[name]
            k["achieved_GBps"] = per_step_bytes[name] / (ms_step_k * 1e-3) / 1e9 if ms_step_k else None
        if name in ("scan_compact_kernel", "node_rows_kernel", "ecc_sums_kernel"):
            lo, med, hi = timing_full[name][2:]
            k["launch_ms_min_median_max"] = [lo, med, hi]      # spread over the timed launches (device state)
        kernels[name] = k
    # the dominant kernel of the frame loop.  (With the build on its own stream the ray-casting kernels run BESIDE pass A and are
    # stretched over it by the sharing -- their event times are not time the step waits for; they stay in "kernels".)
    side = ("projection_kernel", "heavy_kernel", "witness_kernels", "primary_list_kernel", "retry_list_kernel", "project_nodes_kernel",
            "projection_finish_kernels", "amap_build")
    dom = max((n for n in kernels if n in per_step_bytes and n != "ecc_sums_kernel" and not (overlap and n.startswith(side))),
              key=lambda n: kernels[n]["ms_per_step"])
    dk = kernels[dom]
    calls = max(dk["calls_per_step"], 1)
    # HBM traffic of that kernel: only from a rocprofv3 PMC summary of THIS configuration, handed over
    # explicitly (tools/profile_bench.sh writes it: FETCH_SIZE and WRITE_SIZE in separate passes, the
    # gfx950 FETCH correction applied for the streaming kernels); otherwise null
    traffic, traffic_src = tracked_traffic(dom, "ecc" if a.registration else "bench", world)
    roof = {"kernel": dom, "bound": "hbm",
            "achieved": dk["achieved_GBps"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": dk["achieved_GBps"] / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
            "algorithmic_bytes_per_launch": per_step_bytes[dom] / calls,
            "avg_launch_ms": dk["avg_launch_ms"], "launches_per_step": calls}
    roof.update(ECC_SYMBOLS.get(dom, {}))
    if rank == 0:
        # all algorithmic bytes of a step (every kernel that has a per-step figure; the ECC blend label is not counted twice)
        roofline_extras(roof, {k: v for k, v in per_step_bytes.items() if k in kernels and k not in ("ecc_sums_identity", "ecc_sums_general")}, ms_step)
    if overlap:
        roof["note"] = ("default schedule: the ray casting of the projection builds has a high-priority stream of its own and runs beside "
                        "pass A (scan_compact_kernel) and pass B (node_rows_kernel), which share the memory system with it; alone (--serial) "
                        "pass A takes 0.40 ms = 0.66 of peak and pass B 0.36 ms = 0.70, and the step 25-30 % longer")

    sched = ("projection build, then hot-pixel scan + gather kernels per 64-frame sub-batch"
             if (a.two_kernel or a.registration) else
             ("one upsp_pipeline_step call per step: the projection builds on two high-priority streams of the pipeline, used in turn (two "
              "builds in flight; with the candidate-pixel map, the restore of the hot pixels, the finals, the projection's hand-over and the "
              "node -> row sweep), the frame loop's stream carries pass A (scan + compact pixel series of the candidate pixels%s), the "
              "hot-pixel repair and pass B (whole rows) only"
              % (", in two launches" if (lean and not chunked) else "")) if (overlap and lean) else
             "the ray casting of the projection build on a high-priority stream of its own, pass A (scan + compact pixel series of the "
             "candidate pixels) beside it, then pass B (whole rows)" if overlap else
             "projection build, then pass A (scan + compact pixel series) and pass B (whole rows) per <= 1024 frames")
    out = {
        "metric": "frames/s", "value": fps, "unit": "frames/s", "n_gpus": world,
        "steps": a.steps, "warmup": a.warmup, "ms_per_step": ms_step, "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "f32",
        "data": "synthetic", **({"rendezvous_backend": backend} if world > 1 and backend != "nccl" else {}),
        # (set only by the tests: a stand-in RCCL that lets several rank processes share one GPU -- such a line is not a measurement)
        **({"rccl_library": os.environ["UPSP_RCCL_LIBRARY"]} if os.environ.get("UPSP_RCCL_LIBRARY") else {}),
        **({"collectives": "issued through %s in a one-rank group" % ("RCCL" if backend == "nccl" else backend)} if force_coll else {}),
        "config": {"workload": "configs[%d]: %d frames/GPU x %dx%d u16, %d-tri tunnel model (%d nodes), "
                               "raycast+%sprojection" % (2 if a.registration else 1, F, size, size,
                                                         tris.shape[0], N,
                                                         "registration+" if a.registration else ""),
                   "frames_per_gpu": F, "nodes": N, "triangles": int(tris.shape[0]),
                   "active_pixels": n_active, "series_row_pitch": ld if not chunked else None,
                   "frame_content": ("model intensity + noise" if a.plain_frames else
                                     "SURVEY 8(d): model intensity x 24 fiducial discs, background 60, noise 8, "
                                     "<= 3 hot pixels in 1 % of the frames"),
                   "parallelism": "frames sharded x%d" % world, "schedule": sched,
                   **({"exchange": "%d chunks, %s as %s%s" % (K, "active-pixel series" if pixel_wire else "visible rows",
                                                              ("u16 packed to 12 bit" if a.wire12 else "u16") if u16_wire else "f32",
                                                              ("; two exchanges in turn, a step's series finished behind the next step's chunks" if deferred else "") +
                                                              ("" if not pixel_wire else
                                                               "; pass A once for the rank's frames beside the projection build" if px_once and overlap else
                                                               "; pass A per chunk after the projection build"))}
                      if chunked else {})},
        # rays the REFERENCE casts for this camera / the time of a build that casts a tenth of them (the oblique test first, the
        # occluder witness): an equivalence, not a ray rate -- "mrays_per_s" below is the ray caster's own rate
        "mrays_reference_equivalent_per_s": mrays,
        "mrays_reference_equivalent_kind": "rays the reference casts / build time%s" % (
            " of builds timed alone after the steps: in the steps pass A runs beside the build" if overlap else ""),
        "rays_per_step": nrays, "rays_cast_per_step": primary_rays + n_retry_rays,
        "mrays_cast_per_s": (primary_rays + n_retry_rays) / (ray_alone_ms * 1e-3) / 1e6,
        "rays_note": "nodes the oblique test rejects cast no ray in the timed builds (they have no entry whatever "
                     "their rays say); a build in the reference's order gives the same entries: %s" % same_entries,
        # (default schedule: "projection_build" = the build WITH pass A beside it, "frame_loop" = what is left of the loop after
        #  it, pass B and the repair; "projection_build_alone" = the build timed by itself after the timed steps)
        # (one-call step: no event of this file sits between the build and the frame loop any more -- "projection_build" = the sum of
        #  the build's kernels as the library's timers saw them on the side stream, stretched by the passes beside them;
        #  "frame_loop" = the whole step on this process's stream: pass A, the repair, the wait for the build, pass B)
        "breakdown_ms": {"projection_build": (sum(v["ms_per_step"] for n, v in kernels.items() if n.startswith(side)) if lean else ray_ms),
                         "frame_loop": frm_ms, "exchange_finals": float(np.mean(t_xchg)),
                         **({"projection_build_alone": ray_alone_ms} if overlap else {})},
        "breakdown_ms_per_step": {"projection_build": [round(x, 4) for x in t_ray], "frame_loop": [round(x, 4) for x in t_frames],
                                  "exchange_finals": [round(x, 4) for x in t_xchg],
                                  # time the HOST spent issuing each phase (no synchronisation inside a step: the GPU runs behind)
                                  "host_issue": [[round(x, 4) for x in h] for h in host_log]},
        "frame_loop_frames_per_s": F / ((frm_ms_share if overlap else frm_ms) * 1e-3),
        # whole frame loop against HBM: (2 MiB + 4 B x N) per frame + 8 B x N per series launch
        "frame_loop_GBps": (F * (2 * npx + series_esz * series_rows) +
                            (row_launches if streamed else gather_launches) * 8 * N) / ((frm_ms_share if overlap else frm_ms) * 1e-3) / 1e9,
        "roofline": roof,
        "kernels": kernels,
    }
    if chunked:
        out["exchange_self_check"] = xcheck      # every rank: its own frames' columns of its series slice == frame[pix], NaN rows
        out["exchange_finals_check"] = fcheck    # every rank: finals (all-reduced sums) of its nodes == mean of their complete series
    if world > 1 or force_coll:
        cr = D.comm_ranks()
        out["rccl_nranks"] = None if cr is None else cr[1]       # ncclCommCount of the communicator the exchanges ran on
        out["rccl_bound"] = _capi.comm_library()                 # the RCCL build behind upsp_comm_* (the process's own copy when it has one)
        out["ms_per_step_rank_min_max"] = [dt_rank_min / a.steps * 1e3, dt_rank_max / a.steps * 1e3]
    if chunked:
        # what a rank hands to the exchange per step, and how much of it crossed a link in THIS run (nothing in a one-rank
        # group; (N - 1) / N of it at N ranks, one block per xGMI link and chunk)
        wire_bytes = 1.5 if (u16_wire and a.wire12) else series_esz
        xb = exch.exchange_bytes()
        out["exchange_bytes_per_step"] = {
            "travelling_rows": int(series_rows), "what_travels": "active-pixel series" if pixel_wire else "node rows",
            "frames_per_rank": F, "wire_bytes_per_value": wire_bytes,
            "packed_series_bytes_per_rank": int(series_rows * F * wire_bytes),
            "leaves_the_gpu_at_8_ranks": int(series_rows * F * wire_bytes * 7 / 8),
            "sent_to_other_ranks_this_run": None if xb is None else xb[0],
            "transport": ("C ABI upsp_exchange_* over RCCL" if (world > 1 or force_coll) else "C ABI upsp_exchange_*, one rank in process (device copies)")}
    if a.registration:
        out["ecc_iterations_per_frame"] = ecc_iters_per_frame
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
    if world == 1 and not a.registration and not a.no_reraycast:
        out["host_feed"] = host_feed_rate(pipe, frames, N, size, last_pix[0])
        if a.no_cpu_baseline:
            out["pixel_rays"] = pixel_ray_rate(bvh, cd, size)
    plain_default = world == 1 and not a.registration and not a.no_reraycast and not chunked and not a.small
    if plain_default:
        # BASELINE configs[2] beside the headline: per-frame ECC registration in front of the projection -- at configs[2]'s OWN
        # size (10 000 resident frames, 2 steps) when the device has the memory for it (20 GB of frames + 20 GB of series + 2 GB
        # of registration scratch beside what this run holds), else on the 1 000 frames of the headline (5 steps)
        F2 = int(os.environ.get("UPSP_BENCH_CONFIGS2_FRAMES", "10000"))
        free_b = torch.cuda.mem_get_info()[0]
        if F2 > F and F == 1000 and free_b >= 60e9:
            restore_hot()
            frames2 = torch.empty((F2, size, size), dtype=torch.uint16, device="cuda")
            frames2[:F] = frames
            for f0 in range(F, F2, chunk):
                syn.synth_frames_torch(min(chunk, F2 - f0), size, size, first=f0, out=frames2[f0:f0 + chunk], layout=layout, hot=True)
            out["configs2"] = registration_block(bvh, cam, d_nodes, d_nrm, d_tn, frames2, hot_pixel_restorer(frames2), N, size, n_active,
                                                 steps=2, warmup=1)
            del frames2
        else:
            out["configs2"] = registration_block(bvh, cam, d_nodes, d_nrm, d_tn, frames, restore_hot, N, size, n_active,
                                                 steps=5, warmup=2)
            out["configs2"]["note"] = "%.0f GB of HBM free: configs[2] on the headline's %d frames, not its own 10 000" % (free_b / 1e9, F)
    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        with_reg = a.registration or plain_default
        out["cpu_baseline"], ref = cpu_baseline(verts, tris, cd, size, F, sample, registration=with_reg)
        gpix = last_pix[0].cpu().numpy()
        checks = {
            "projection_pix_full_model": bool(np.array_equal(gpix, ref["pix"])),
            "reference_ray_count": bool(nrays == ref["nrays"]),
            "oblique_test_first_same_entries": same_entries,
        }
        if not a.registration:
            # parity of THIS run against the oracle: the projection of the full 1 M-triangle model, and the
            # frame loop on the sample the CPU just processed (same bits in: the frames as generated)
            d_sample = torch.as_tensor(sample.view(np.int16)).view(torch.uint16).cuda()
            p2 = engine.FramePipeline(1, size, size, N, fused_scan=2 if a.two_kernel else 0)
            rt = torch.empty((N, engine.series_ld(n_sample, whole_rows=True)), dtype=torch.float32, device="cuda")[:, :n_sample]
            if lean:
                # through the same entry point as the timed steps: ONE upsp_pipeline_step call (its own projection build on the side
                # stream, pass A on the candidate map in two launches, the projection handed over in the pipeline's buffer, the
                # node -> row sweep, whole-line rows)
                p2.set_scan_split(scan_split_on)
                p2.set_row_padding(row_padding)
                p2.step(bvh, cam, d_nodes, d_nrm, d_tn, d_sample, rows_t=rt, first_frame=0)
                p2.step_finish()
                checks["step_call_same_projection"] = bool(torch.equal(p2.current_projection(0), last_pix[0]))
            else:
                p2.set_projection(0, last_pix[0])
                p2.process(d_sample, first_frame=0, rows_t=rt, want_rows=False)
            gs, gss = [x.cpu().numpy() for x in p2.accumulators()]
            ok = ~np.isnan(ref["sum"])
            checks.update({
                "series_rows_8_frames": bool(np.array_equal(rt[:, :8].cpu().numpy().T.view(np.int32), ref["rows8"].view(np.int32))),
                "repaired_frames": bool(np.array_equal(d_sample.cpu().view(torch.int16).numpy().view(np.uint16), ref["frames_fixed"])),
                "accumulators_%d_frames" % n_sample: bool(np.array_equal(np.isnan(gs), ~ok) and np.array_equal(gs[ok], ref["sum"][ok])
                                                       and np.array_equal(gss[ok], ref["sumsq"][ok])),
            })
            if chunked:   # the series as it came out of the (chunked, packed, u16) exchange
                checks["exchange_series_8_frames"] = bool(np.array_equal(
                    ex_state["last_done"].out[:, :8].cpu().numpy().T.view(np.int32), ref["rows8"].view(np.int32)))
            if not a.no_reraycast:
                out["pixel_rays"] = pixel_ray_rate(bvh, cd, size, check_with=ref["obv"])
                checks["pixel_rays_closest_hit_sample"] = out["pixel_rays"]["parity"]
        if with_reg:
            # configs[2]: warps, iteration counts and registered rows of the frames the CPU baseline registered
            rchecks, rinfo = registration_parity(ref, sample, last_pix[0], N, size)
            checks.update({"registration_" + k: v for k, v in rchecks.items()})
            tgt = out if a.registration else out["configs2"]
            tgt["registration_parity"] = rinfo
            if not a.registration:
                # CPU baseline of configs[2]: the same oracle loop with register_pixel per frame
                pf = ref["loop_per_frame"] + ref["reg"]["seconds_per_frame"] / ref["cores"]
                Fc2 = int(out["configs2"]["frames"])
                out["configs2"]["cpu_baseline"] = {
                    "value": Fc2 / (ref["t_proj"] + ref["t_fixed"] + pf * Fc2), "unit": "frames/s", "cores": ref["cores"], "kind": "port",
                    "sample": "oracle/ (C, %d threads, one frame per thread): fix_hot_pixels + register_pixel + project_frame on %d frames "
                              "(%.2f s per frame and thread) + the plain loop's per-frame cost; extrapolated to %d frames"
                              % (ref["cores"], len(ref["reg"]["fixed"]), ref["reg"]["seconds_per_frame"], Fc2)}
                # the headline's CPU baseline is the PLAIN loop (configs[1]): take the registration cost out again
                cb = out["cpu_baseline"]
                cb["value"] = F / (ref["t_proj"] + ref["t_fixed"] + ref["loop_per_frame"] * F)
                cb["frame_loop_frames_per_s"] = 1.0 / ref["loop_per_frame"]
                cb["sample"] = cb["sample"].split(" + fix_hot_pixels, register_pixel")[0] + "; extrapolated to the %d-frame step" % F
        if plain_default:
            # the plain ray caster once more on a scene where every pixel's ray enters the tree: a 1 M-triangle sphere
            # that fills the frame (the tunnel model covers 8 % of the pixels)
            fv, ft = syn.cube_sphere(289, 6.0)
            fs9, _ = syn.soup(fv, ft)
            fcd = syn.pinhole_camera(size, size, center=(0, 0, 20), half_extent=6.0, fill=0.95)
            fbvh = engine.BVH(fs9)
            out["pixel_rays_fill"] = pixel_ray_rate(fbvh, fcd, size, check_with=orc_bvh(fs9))
            out["pixel_rays_fill"]["scene"] = "cube sphere, %d triangles, filling the frame" % ft.shape[0]
            checks["pixel_rays_fill_closest_hit_sample"] = out["pixel_rays_fill"]["parity"]
            fbvh.close()
        out["parity_checked"] = all(checks.values())
        out["parity"] = checks
        if not out["parity_checked"]:
            print(json.dumps(out), flush=True)
            raise SystemExit("bench.py: GPU results differ from the oracle: %r" % (checks,))
    elif rank == 0:
        out["cpu_baseline"] = None
    if rank == 0:
        # the key named like BASELINE.json's ray metric: closest-hit rays that ENTER the tree per second, on the scene where every
        # pixel's ray does (pixel_rays_fill) when this run measured it, on the tunnel model otherwise
        pr = out.get("pixel_rays_fill") or out.get("pixel_rays")
        out["mrays_per_s"] = pr.get("mrays_entered_per_s") if pr else None
        out["mrays_per_s_kind"] = (None if not pr else "closest-hit rays entering the tree per second, one ray per pixel, %s"
                                   % ("frame-filling 1 M-triangle sphere (pixel_rays_fill)" if "pixel_rays_fill" in out else "tunnel model (pixel_rays)"))
        hf, c2 = out.get("host_feed"), out.get("configs2")
        out["summary"] = "%.0f k frames/s with the frames resident in HBM (%d GPU%s)%s%s%s" % (
            fps / 1e3, world, "" if world == 1 else "s",
            "; %.1f k frames/s when they arrive from host memory (PCIe-bound, never `value`)" % (hf["frames_per_s"] / 1e3) if hf and hf.get("frames_per_s") else "",
            "; configs[2] (ECC registration, parity unpinned) %.0f k frames/s" % (c2["value"] / 1e3) if c2 else "",
            "; %.2f G rays/s entering the tree" % (out["mrays_per_s"] / 1e3) if out.get("mrays_per_s") else "")
    # BASELINE configs[3] at its own size, inside the N > 1 line (every rank takes part; rank 0 holds the line)
    if (not nested and world > 1 and chunked and pixel_wire and not a.config3_share and not a.registration and not a.chunk_scan and
            not a.sync_exchange and (not a.small or os.environ.get("UPSP_BENCH_CONFIGS3_FRAMES"))):
        blk3 = configs3_block(a, world, N, size * size)
        if rank == 0:
            out["configs3"] = blk3
    if rank == 0 and not nested:
        print(json.dumps(out), flush=True)
    if exch is not None:
        for x in exchs:
            x.verify()                  # (the timer-on steps made assume_same claims too)
            x.close()
    if nested:
        if xcheck is False:
            raise SystemExit("bench.py (configs[3] block): the series out of the exchange differ from this rank's frames on at least one rank")
        return out
    if world > 1 or force_coll:
        D.shutdown()
    if xcheck is False:
        raise SystemExit("bench.py: the series out of the exchange differ from this rank's frames on at least one rank")
    return out


if __name__ == "__main__":
    main()
