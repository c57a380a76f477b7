"""Phase 1 of psp_process on the GPU engine (cpp/exec/psp_process.cpp:1438-2040).

`Phase1` mirrors the reference's control flow for the hot path and nothing else:

    createBVH -> per camera create_projection_mat -> adjust_projection_for_weights ->
    identify_skipped_nodes -> first-frame solution sol1 -> frame loop -> reductions ->
    finals (avg, rms, intensity_ratio_0, coverage) -> node-major time series

Inputs are flat arrays (what the reference's model / calibration / video readers produce);
outputs are tensors plus, optionally, the reference's flat files (raw little-endian f32,
no header; cpp/exec/psp_process.cpp:524-540, docs/sphinx/file-formats.rst:821-894).

Multi-GPU: construct under torch.distributed (one process per GPU).  Frames are sharded
with apportion(); all cameras of a frame stay on one GPU so the per-node fusion is local.
"""
import os

import numpy as np
import torch
import torch.distributed as dist

from . import _capi, distributed as D, engine


class Phase1:
    def __init__(self, tris9, tri_nodes, nodes, normals, cameras, image_size, oblique_angle=70.0,
                 overlap="average_view", datanode=None, registration=False, interp=1,
                 filter=None, filter_size=1, patches=None, nframes_total=None, targets=None,
                 first_frames=None, bit_depth=12, bound_pts=2, buffer_pts=1, target_diam_sf=1.2,
                 overlap_src=None, count_rays=False):
        """cameras: list of dict(K, dist, R, t); image_size = (width, height).

        patches: per camera list of dict(bx, by, ix, iy) (PatchClusters tables), or give
        `targets` (one target file per camera) + `first_frames` (raw u16 [H,W] frame 1 per
        camera) to run the phase-0 set-up here (InitializeImagePatches, :2088-2182).
        overlap_src: int32 [N] source map of a structured (PLOT3D) model's zone overlaps
        (grids.P3DModel.overlap_source(); model.adjust_solution, :1833-1835, :1938-1940, :1975-1977).
        count_rays: `self.nrays` is the number of rays the REFERENCE casts for these cameras (every in-frame node's
        primary ray + its retries: the build then casts them all, in the reference's order); by default it is the
        number of rays this build really cast (nodes the oblique test rejects cast none, most retries are decided
        by the occluder witness) -- `self.nrays_kind` says which."""
        self.width, self.height = image_size
        self.ncams = len(cameras)
        self.nnodes = int(np.asarray(nodes).reshape(-1, 3).shape[0])
        self.bvh = engine.BVH(tris9)                                  # createBVH :44-53
        self.d_nodes = torch.as_tensor(np.ascontiguousarray(nodes, np.float32)).cuda()
        self.d_normals = torch.as_tensor(np.ascontiguousarray(normals, np.float32)).cuda()
        self.d_tri_nodes = torch.as_tensor(np.ascontiguousarray(tri_nodes, np.int32)).cuda()
        self.bvh.set_tri_nodes(self.d_tri_nodes, self.nnodes)        # createBVH(model, triNodes)
        self.cams = [_capi.make_camera(c["K"], c["dist"], c["R"], c["t"], self.width, self.height)
                     for c in cameras]
        self.centers = np.array([engine.camera_center(c) for c in self.cams])
        self.nrays = 0
        self.nrays_kind = "rays the reference casts" if count_rays else "rays cast"
        pix, self.uv, self.nodecount = [], [], []
        for cam in self.cams:                                          # :1597-1622
            # (count_rays: also report the number of rays the reference would cast -- which means casting them all;
            #  otherwise nodes the oblique test rejects cast none, engine.build_projection)
            p = engine.build_projection(self.bvh, cam, self.d_nodes, self.d_normals,
                                        self.d_tri_nodes, oblique_angle, datanode=datanode, nodecount=True,
                                        counts=count_rays)
            pix.append(p["pix"])
            self.uv.append(p["uv"])
            self.nodecount.append(p["nodecount"])      # u8 [H, W], saturating (psp_process.cpp:335-347)
            self.nrays += p["nrays"] if count_rays else engine.projection_counts(self.bvh)["nrays"]
        self.pix = torch.stack(pix)
        # :1632-1640 ; a single camera needs no weights
        self.weight = (engine.projection_weights(self.pix, self.d_nodes, self.d_normals,
                                                 self.centers, overlap)
                       if self.ncams > 1 else None)
        self.skipped, self.nskipped = engine.skipped_nodes(self.pix)   # :1644
        self.visible_targets = None
        if patches is None and targets is not None:
            from . import patch_setup
            patches, self.visible_targets = [], []
            for c, cam in enumerate(self.cams):
                fr = first_frames[c]
                fr = fr.cpu().numpy() if isinstance(fr, torch.Tensor) else np.asarray(fr)
                tab, vis, _ = patch_setup.initialize_image_patches(
                    self.bvh, cam, image_size, targets[c], fr.reshape(self.height, self.width),
                    self.d_nodes, normals, oblique_angle=oblique_angle, bit_depth=bit_depth,
                    bound_pts=bound_pts, buffer_pts=buffer_pts, target_diam_sf=target_diam_sf)
                patches.append(tab)
                self.visible_targets.append(vis)
        opts = dict(registration=int(bool(registration)), interp=int(interp),
                    patch=int(patches is not None))
        if filter:
            opts.update(filter={"gaussian": 1, "box": 2}[filter], filter_size=int(filter_size))
        self.pipe = engine.FramePipeline(self.ncams, self.width, self.height, self.nnodes, **opts)
        for c in range(self.ncams):
            self.pipe.set_projection(c, self.pix[c], None if self.weight is None else self.weight[c])
            if patches is not None:
                self.pipe.set_patches(c, patches[c])
        self.pipe.set_skipped(self.skipped)
        self.overlap_src = None
        if overlap_src is not None:
            self.overlap_src = torch.as_tensor(np.ascontiguousarray(overlap_src, np.int32)).cuda()
            self.pipe.set_overlap_source(self.overlap_src)
        self.registration = bool(registration)
        self._patches = patches
        self.sol1 = None
        self.frames_done = 0

    # -- first frame (psp_process.cpp:1655-1713) ------------------------------------
    def set_first_frames(self, first_frames):
        """first_frames: list of RAW u16 [H,W] tensors (frame 1 of every camera).

        * ECC template of the frame loop = raw first frame as f32, without hot-pixel
          repair (elems.first_frames, psp_process.cpp:2057-2058);
        * sol1 = the hot-pixel-repaired first frame (first_frames_raw, :880-884) registered
          against its own f32 copy (first_frames_32f, :1676), patched, filtered, projected."""
        raws = [f.to(device="cuda").reshape(self.height, self.width).contiguous() for f in first_frames]
        for c, r in enumerate(raws):
            self.pipe.set_reference(c, r.to(torch.float32))
        fixed = [r.clone().reshape(1, self.height, self.width) for r in raws]
        for f in fixed:
            engine.fix_hot_pixels(f)
        o = self.pipe.opts
        tmp = engine.FramePipeline(self.ncams, self.width, self.height, self.nnodes,
                                   hot_enable=0, registration=o.registration, interp=o.interp,
                                   filter=o.filter, filter_size=o.filter_size, patch=o.patch)
        for c in range(self.ncams):
            tmp.set_projection(c, self.pix[c], None if self.weight is None else self.weight[c])
            tmp.set_reference(c, fixed[c][0].to(torch.float32))
            if self._patches is not None:
                tmp.set_patches(c, self._patches[c])
        tmp.set_skipped(self.skipped)
        if self.overlap_src is not None:
            tmp.set_overlap_source(self.overlap_src)                  # model.adjust_solution(sol1), :1713
        frames = fixed
        self.sol1 = tmp.process(frames, first_frame=1)[0].clone()
        tmp.close()
        return self.sol1

    # -- frame loop (psp_process.cpp:1743-1851) --------------------------------------
    def process(self, frames, first_frame, rows_t=None, col0=0, want_rows=False):
        """frames: list (per camera) of u16 [F,H,W] tensors, global index of frame 0 =
        first_frame.  rows_t: optional [N, >=col0+F] node-major output."""
        rows = self.pipe.process(frames, first_frame=first_frame, rows_t=rows_t, col0=col0,
                                 want_rows=want_rows)
        self.frames_done += frames[0].shape[0]
        return rows

    # -- N > 1: which form of the series travels (psp_process.cpp:707-771) ----------------
    def pixel_wire(self, shard):
        """True when the time-series exchange can carry the ACTIVE PIXELS' u16 series instead of f32 node rows: one
        camera, no weights, no float image stage, no overlap map -- every series value is then the integer value of a
        pixel, a third (and less) of the bytes on the links, and the owner of a node runs pass B.
        UPSP_ROW_WIRE=1 forces the node rows."""
        return (shard.world > 1 and self.ncams == 1 and self.weight is None and not self.registration
                and self._patches is None and not self.pipe.opts.filter and self.overlap_src is None
                and not os.environ.get("UPSP_ROW_WIRE"))

    def frame_loop_pixel_wire(self, shard, read_chunk, chunk=256, progress=None):
        """This rank's frame loop with the pixel series on the wire: pass A per chunk (hot-pixel repair included), the
        chunk's series sent while the next chunk is read and scanned, pass B over all frames on the owner of the nodes.
        read_chunk(c0, n) -> list with the camera's u16 [n, H, W] device tensor of this rank's frames c0 .. c0 + n.
        Returns this rank's [nodes_r, F] f32 slice; the accumulators hold this rank's share of the sums (finalize()
        all-reduces them as always)."""
        # K is the same on every rank and no chunk of any rank is longer than `chunk` frames -- what read_chunk's
        # buffers hold (psp_process.py sizes its feed slots with it) -- nor than one pass A group of this pipeline
        limit = min(int(chunk), self.pipe.series_frames_max())
        K = D.chunk_count(shard.frame_count, limit)
        ex = D.TimeSeriesExchange(shard, K)
        assert max(max(e) for _, e in ex.chunks) <= limit
        tab = self.pipe.pixel_series(None)                             # the active-pixel map alone
        ex.set_pixels(tab["node_k"], self.skipped)
        for k in range(K):
            c0, fc = ex.my_chunk(k)
            if fc:
                batch = read_chunk(c0, fc)
                ex.submit_pixels(self.pipe.pixel_series(batch[0].contiguous()))
                self.frames_done += fc
            else:
                ex.submit_pixels(tab)                                  # (a rank with fewer chunks still takes part)
            if progress:
                progress(c0)
        s, ss = self.pipe.accumulators()
        series = ex.finish_pixels(s, ss)
        torch.cuda.current_stream().synchronize()
        ex.close()
        return series

    # -- reductions + finals (psp_process.cpp:1866-1979) -----------------------------
    def finalize(self, nframes_total):
        s, ss = self.pipe.accumulators()
        D.allreduce_sums(s, ss)                                        # MPI_Reduce + MPI_Bcast
        avg, rms = self.pipe.finalize(nframes_total)
        if self.overlap_src is not None:                               # :1937-1940
            idx = self.overlap_src.long()
            avg, rms = avg[idx], rms[idx]
        out = dict(avg=avg, rms=rms)
        if self.sol1 is not None:
            out["ratio_0"] = avg / self.sol1 - 1.0                     # :1948-1950
        # coverage = sum_c project(ones) (:1955-1975)
        ones = torch.ones((self.height, self.width), dtype=torch.float32, device="cuda")
        cov = None
        for c in range(self.ncams):
            w = None if self.weight is None else self.weight[c].contiguous()
            ind = engine.project_frame(ones, self.pix[c].contiguous(), w)
            cov = ind if cov is None else cov + ind
        if self.overlap_src is not None:                               # :1975-1977
            cov = cov[self.overlap_src.long()]
        out["coverage"] = cov
        return out

    @staticmethod
    def dump_vv(path, v, maxels=1000):
        """Regression slices vv-*.dat (psp_process.cpp:1984-2016)."""
        v = np.asarray(v, dtype=np.float32)
        step = 1 if v.size < maxels else v.size // maxels
        np.ascontiguousarray(v[::step][:maxels]).tofile(path)

    def write_outputs(self, out_dir, finals, series=None, node_start=0):
        """Flat files of phase 1.  `series`: this rank's [nodes_r, F] slice, written at its
        byte offset into the shared intensity_transpose file like the reference's pwrite
        (:958-963)."""
        os.makedirs(out_dir, exist_ok=True)
        rank = dist.get_rank() if dist.is_initialized() else 0
        if rank == 0:
            for name, key in (("intensity_avg", "avg"), ("intensity_rms", "rms"),
                              ("intensity_ratio_0", "ratio_0"), ("coverage", "coverage")):
                if key in finals:
                    finals[key].cpu().numpy().astype("<f4").tofile(os.path.join(out_dir, name))
            for c in range(self.ncams):
                self.uv[c].cpu().numpy().astype("<f4").tofile(
                    os.path.join(out_dir, "cam%02d-uv" % (c + 1)))
                write_nodecount_png(os.path.join(out_dir, "cam%02d-nodecount.png" % (c + 1)),
                                    self.nodecount[c].cpu().numpy())
            self.dump_vv(os.path.join(out_dir, "vv-int-rms.dat"), finals["rms"].cpu().numpy())
            self.dump_vv(os.path.join(out_dir, "vv-int-avg.dat"), finals["avg"].cpu().numpy())
            self.dump_vv(os.path.join(out_dir, "vv-int-coverage.dat"), finals["coverage"].cpu().numpy())
            if self.sol1 is not None:
                self.dump_vv(os.path.join(out_dir, "vv-int-sample1.dat"), finals["ratio_0"].cpu().numpy())
        if series is not None:
            path = os.path.join(out_dir, "intensity_transpose")
            nf = series.shape[1]
            data = series.cpu().numpy().astype("<f4")
            fd = D.create_shared_file(path, self.nnodes * nf * 4)
            try:
                os.pwrite(fd, data.tobytes(), nf * node_start * 4)
            finally:
                os.close(fd)

    def close(self):
        self.pipe.close()
        self.bvh.close()


def write_nodecount_png(path, counts):
    """camNN-nodecount.png (psp_process.cpp:1608-1614): the nodes-per-pixel image through the colour map
    of upsp::nodes_per_pixel_colormap (cpp/utils/cv_extras.cpp:277-290: 0 black, 1 green, 2 yellow, 3 orange,
    4 light orange, >= 5 white).  The reference adds a colour bar with text labels to the right of the image
    (cv::putText); this writer stores the colour-mapped image alone (8-bit RGB PNG, zlib only)."""
    import struct
    import zlib
    lut = np.full((256, 3), 255, np.uint8)
    lut[:5] = [(0, 0, 0), (0, 255, 0), (255, 255, 0), (255, 153, 51), (255, 204, 153)]     # RGB of the BGR table
    img = lut[np.asarray(counts, dtype=np.uint8)]
    h, w = img.shape[:2]
    raw = np.concatenate([np.zeros((h, 1), np.uint8), img.reshape(h, w * 3)], axis=1).tobytes()   # filter 0 per row

    def chunk(tag, data):
        return struct.pack(">I", len(data)) + tag + data + struct.pack(">I", zlib.crc32(tag + data) & 0xFFFFFFFF)
    with open(path, "wb") as f:
        f.write(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 2, 0, 0, 0))
                + chunk(b"IDAT", zlib.compress(raw, 6)) + chunk(b"IEND", b""))


def run_phase1(job, frames_per_cam, nframes_total=None, out_dir=None, chunk=256):
    """Drive a whole phase 1 on this rank's share of the frames.

    frames_per_cam: list (per camera) of u16 [F_total, H, W] arrays / tensors holding ALL
    frames (every rank slices its own share with apportion()).  Returns (finals, series)."""
    F = int(frames_per_cam[0].shape[0]) if nframes_total is None else int(nframes_total)
    shard = D.Shard(F, job.nnodes)
    f0, nf = shard.my_frames
    job.set_first_frames([torch.as_tensor(np.asarray(fr[0])) if not isinstance(fr, torch.Tensor) else fr[0]
                          for fr in frames_per_cam])
    def to_dev(c0, n):
        batch = []
        for fr in frames_per_cam:
            b = fr[f0 + c0:f0 + c0 + n]
            if not isinstance(b, torch.Tensor):
                b = torch.as_tensor(np.ascontiguousarray(b))
            batch.append(b.to("cuda").contiguous())
        return batch
    if job.pixel_wire(shard):
        series = job.frame_loop_pixel_wire(shard, to_dev, chunk)
        finals = job.finalize(F)
        if out_dir:
            job.write_outputs(out_dir, finals, series, node_start=shard.my_nodes[0])
        return finals, series
    rows_t = torch.empty((job.nnodes, engine.series_ld(max(nf, 1))), dtype=torch.float32,
                         device="cuda")[:, :max(nf, 1)]
    job.pipe.set_row_padding(True)          # (columns nf .. series_ld(nf) of that allocation are padding)
    for c0 in range(0, nf, chunk):
        n = min(chunk, nf - c0)
        batch = []
        for fr in frames_per_cam:
            b = fr[f0 + c0:f0 + c0 + n]
            if not isinstance(b, torch.Tensor):
                b = torch.as_tensor(np.ascontiguousarray(b))
            batch.append(b.to("cuda").contiguous())
        job.process(batch, first_frame=f0 + c0, rows_t=rows_t, col0=c0)
    finals = job.finalize(F)
    series = D.exchange_time_series(rows_t[:, :nf], shard)
    if out_dir:
        job.write_outputs(out_dir, finals, series, node_start=shard.my_nodes[0])
    return finals, series
