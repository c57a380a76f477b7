"""Device-resident wrappers over the C ABI (include/upsp_gpu.h).

torch is plumbing here: it owns the HBM buffers (tensors) and the HIP stream; every
computation happens inside libupsp_gpu.so.  All functions take / return CUDA(HIP)
tensors and launch on torch's current stream.
"""
import ctypes as C
import math

import numpy as np
import torch

from . import _capi
from ._capi import check, lib


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _ptr(t):
    return None if t is None else C.c_void_p(t.data_ptr())


def _dev(x, dtype):
    """numpy / tensor -> contiguous device tensor of dtype."""
    if isinstance(x, torch.Tensor):
        return x.to(device="cuda", dtype=dtype).contiguous()
    return torch.as_tensor(np.ascontiguousarray(x), dtype=dtype).to("cuda").contiguous()


class BVH:
    """rt::BVH on the GPU (reference: cpp/raycast/pspRT.cpp:313-431).

    Parameters: triangle soup, 9 floats per triangle (rt::CreateBVH(raw, 3))."""

    def __init__(self, tris9):
        tris9 = np.ascontiguousarray(tris9, dtype=np.float32).reshape(-1)
        if tris9.size % 9:
            raise ValueError("triangle soup must hold 9 floats per triangle")
        h = C.c_void_p()
        check(lib().upsp_bvh_create(tris9.ctypes.data_as(C.c_void_p), tris9.size // 9, C.byref(h)))
        self._h = h
        self._destroy = lib().upsp_bvh_destroy      # bound now: module globals vanish at exit
        self.ntris = tris9.size // 9

    def share(self):
        """A second handle on the same tree (and the adjacency set so far) with query scratch of its own: builds / batches on the two
        may run at the same time on different streams (upsp_bvh_share).  Keeps this object alive."""
        other = BVH.__new__(BVH)
        h = C.c_void_p()
        check(lib().upsp_bvh_share(self._h, C.byref(h)))
        other._h, other._destroy, other.ntris, other._owner = h, self._destroy, self.ntris, self
        other._tri_nodes = getattr(self, "_tri_nodes", None)
        return other

    def close(self):
        if getattr(self, "_h", None):
            self._destroy(self._h)
            self._h = None

    __del__ = close

    @property
    def handle(self):
        return self._h

    @property
    def info(self):
        i = _capi.BvhInfo()
        check(lib().upsp_bvh_get_info(self._h, C.byref(i)))
        return dict(ntris=i.ntris, n_ref_nodes=i.n_ref_nodes, n_gpu_nodes=i.n_gpu_nodes,
                    depth=i.depth, max_leaf=i.max_leaf, device_bytes=i.device_bytes,
                    build_seconds=i.build_seconds, bounds_min=list(i.bounds_min),
                    bounds_max=list(i.bounds_max))

    def set_tri_nodes(self, tri_nodes, nnodes):
        """Triangle -> node ids of createBVH(model, triNodes) (psp_process.cpp:44-53): device int32
        tensor [3T].  Speeds up build_projection calls that pass the SAME tensor."""
        assert tri_nodes.is_cuda and tri_nodes.dtype == torch.int32 and tri_nodes.is_contiguous()
        check(lib().upsp_bvh_set_tri_nodes(self._h, _ptr(tri_nodes), int(nnodes), _stream()))
        self._tri_nodes = tri_nodes       # keep the buffer (and its address) alive

    def enable_stats(self, on=True):
        check(lib().upsp_bvh_enable_stats(self._h, int(bool(on))))

    def check(self):
        """Waits for the current stream; raises (UPSP_ERR_INTERNAL) if a walk queued on this BVH since the last
        check ran past its round cap -- a broken tree is an error, never a wedged device (upsp_bvh_check)."""
        check(lib().upsp_bvh_check(self._h, _stream()))

    def last_stats(self):
        a, b, c = C.c_uint64(), C.c_uint64(), C.c_uint64()
        check(lib().upsp_bvh_last_stats(self._h, C.byref(a), C.byref(b), C.byref(c)))
        return dict(nodes=a.value, tris=b.value, rays=c.value)

    def last_filter_stats(self):
        """dict(boxes, undecided): box tests of the last launch the slab filter saw / left to the mirrored filter (stats on)."""
        a, b = C.c_uint64(0), C.c_uint64(0)
        check(lib().upsp_bvh_last_filter_stats(self.handle, C.byref(a), C.byref(b)))
        return dict(boxes=int(a.value), undecided=int(b.value))

    @staticmethod
    def _rays(org, dirs):
        dirs = _dev(dirs, torch.float32).reshape(-1, 3)
        org = _dev(org, torch.float32)
        n = dirs.shape[0]
        if org.numel() == 3:
            stride = 0
        else:
            org = org.reshape(-1, 3)
            if org.shape[0] != n:
                raise ValueError("origins must be (3,) or (N,3)")
            stride = 3
        return org, stride, dirs, n

    def intersect(self, org, dirs, want=("hit", "t", "prim", "uvw", "pos", "nrm")):
        """Closest hit of N rays (rt::BVH::intersect semantics). Returns dict of tensors."""
        org, stride, dirs, n = self._rays(org, dirs)
        out = {}
        h = _capi.Hits()
        shapes = dict(hit=((n,), torch.uint8), t=((n,), torch.float32), prim=((n,), torch.int32),
                      uvw=((n, 3), torch.float32), pos=((n, 3), torch.float32),
                      nrm=((n, 3), torch.float32))
        for k in want:
            shp, dt = shapes[k]
            out[k] = torch.empty(shp, dtype=dt, device="cuda")
            setattr(h, k, out[k].data_ptr())
        check(lib().upsp_bvh_intersect(self._h, _ptr(org), stride, _ptr(dirs), n, C.byref(h),
                                       _stream()))
        if "hit" in out:
            out["hit"] = out["hit"].bool()
        return out

    def occluded(self, org, dirs):
        """Boolean return value of rt::BVH::intersect for N rays (any hit)."""
        org, stride, dirs, n = self._rays(org, dirs)
        hit = torch.empty(n, dtype=torch.uint8, device="cuda")
        check(lib().upsp_bvh_occluded(self._h, _ptr(org), stride, _ptr(dirs), n, _ptr(hit),
                                      _stream()))
        return hit.bool()


def oblique_threshold(oblique_angle_deg):
    """deg2_rad(180. - oblique_angle) narrowed to float (psp_process.cpp:1602)."""
    return float(np.float32((180.0 - float(np.float32(oblique_angle_deg))) * 3.141592653589793 / 180.0))


def camera_center(cam):
    c = (C.c_double * 3)()
    check(lib().upsp_camera_center(C.byref(cam), c))
    return np.array(list(c))


def project_points(cam, xyz):
    xyz = np.ascontiguousarray(xyz, dtype=np.float32).reshape(-1, 3)
    uv = np.zeros((xyz.shape[0], 2), np.float32)
    check(lib().upsp_project_points_host(C.byref(cam), xyz.ctypes.data_as(C.c_void_p),
                                         xyz.shape[0], uv.ctypes.data_as(C.c_void_p)))
    return uv


def projection_counts(bvh):
    """Counters of the most recent build_projection on this BVH (waits for the stream):
    dict(nrays = rays by the reference's sequential count, primary_rays, retry_nodes) -- of the nodes that
    were not dropped by the early oblique test when the build ran with counts=False."""
    a, b, c = C.c_uint64(0), C.c_uint64(0), C.c_uint64(0)
    check(lib().upsp_projection_fetch_counts(bvh.handle, C.byref(a), C.byref(b), C.byref(c), _stream()))
    return dict(nrays=int(a.value), primary_rays=int(b.value), retry_nodes=int(c.value))


def build_projection(bvh, cam, nodes, normals, tri_nodes, oblique_angle=70.0, datanode=None,
                     nodecount=False, counts=True, pix_out=None):
    """create_projection_mat (psp_process.cpp:167-355) for one camera.

    Returns dict(pix int32[N] (-1 = no entry), uv f32[2N], nrays, nodecount u8[H,W] | None).
    counts=True: nrays = the number of rays the reference casts; they are all cast (the call waits for the device).
    counts=False: no host synchronisation, and the oblique test runs BEFORE the rays: nodes it rejects (no entry
    whatever their rays say) cast none -- same pix / uv / nodecount, about a third of the rays on a closed body
    (include/upsp_gpu.h, upsp_projection_build).  The counters of that build stay on the device
    (projection_counts).
    pix_out: int32 [N] device tensor to build into (FramePipeline.projection_target(): no copy at set_projection)."""
    nodes = _dev(nodes, torch.float32).reshape(-1, 3)
    normals = _dev(normals, torch.float32).reshape(-1, 3)
    tri_nodes = _dev(tri_nodes, torch.int32).reshape(-1)
    if tri_nodes.numel() != 3 * bvh.ntris:
        raise ValueError("tri_nodes must hold 3 node ids per triangle of the BVH")
    n = nodes.shape[0]
    dn = None if datanode is None else _dev(datanode, torch.uint8).reshape(-1)
    if pix_out is not None:
        assert pix_out.is_cuda and pix_out.dtype == torch.int32 and pix_out.numel() == n and pix_out.is_contiguous()
    pix = torch.empty(n, dtype=torch.int32, device="cuda") if pix_out is None else pix_out
    uv = torch.empty(2 * n, dtype=torch.float32, device="cuda")
    cnt = torch.empty((cam.height, cam.width), dtype=torch.uint8, device="cuda") if nodecount else None
    nrays = C.c_uint64(0)
    check(lib().upsp_projection_build(bvh.handle, C.byref(cam), _ptr(nodes), _ptr(normals),
                                      _ptr(dn), _ptr(tri_nodes), n,
                                      C.c_float(oblique_threshold(oblique_angle)), _ptr(pix),
                                      _ptr(uv), _ptr(cnt), C.byref(nrays) if counts else None, _stream()))
    if not counts:
        return dict(pix=pix, uv=uv, nodecount=cnt)
    pr, rn = C.c_uint64(0), C.c_uint64(0)
    check(lib().upsp_projection_last_counts(bvh.handle, C.byref(pr), C.byref(rn)))
    return dict(pix=pix, uv=uv, nrays=int(nrays.value), nodecount=cnt,
                primary_rays=int(pr.value), retry_nodes=int(rn.value))


def candidate_pixels(cam, nodes, datanode=None, normals=None, oblique_angle_deg=None):
    """Step 1 of create_projection_mat alone: int32 [N], the pixel every in-frame node would be stored at
    (-1 otherwise) -- a superset of any projection of this camera (FramePipeline.set_active_hint).  With normals and the
    oblique angle of the build: only the nodes that pass the oblique test (the ones that cast a primary ray) -- still a
    superset of the projection built with that angle, a third of the pixels on a closed body."""
    nodes = _dev(nodes, torch.float32).reshape(-1, 3)
    dn = None if datanode is None else _dev(datanode, torch.uint8).reshape(-1)
    pix = torch.empty(nodes.shape[0], dtype=torch.int32, device="cuda")
    if normals is not None:
        nrm = _dev(normals, torch.float32).reshape(-1, 3)
        assert nrm.shape[0] == nodes.shape[0] and oblique_angle_deg is not None
        check(lib().upsp_projection_candidate_pixels_oblique(C.byref(cam), _ptr(nodes), _ptr(nrm), _ptr(dn), nodes.shape[0],
                                                             C.c_float(oblique_threshold(oblique_angle_deg)), _ptr(pix), _stream()))
        return pix
    check(lib().upsp_projection_candidate_pixels(C.byref(cam), _ptr(nodes), _ptr(dn), nodes.shape[0], _ptr(pix), _stream()))
    return pix


def projection_weights(pix, nodes, normals, centers, mode="average_view"):
    """adjust_projection_for_weights (projection.ipp:911-1078).  pix: [ncams, N] int32.
    Returns weight [ncams, N] f32 (1 where a single camera sees the node)."""
    pix = _dev(pix, torch.int32)
    ncams, n = pix.shape
    w = torch.ones((ncams, n), dtype=torch.float32, device="cuda")
    centers = np.ascontiguousarray(centers, dtype=np.float64).reshape(ncams, 3)
    m = {"best_view": 0, "average_view": 1}[mode]
    d_nodes = _dev(nodes, torch.float32)      # keep both alive until the launch is queued
    d_normals = _dev(normals, torch.float32)
    check(lib().upsp_projection_weights(ncams, n, _ptr(pix), _ptr(w), _ptr(d_nodes), _ptr(d_normals),
                                        centers.ctypes.data_as(C.c_void_p), m, _stream()))
    return w


def skipped_nodes(pix, want_count=True, as_bool=True):
    """identify_skipped_nodes (projection.ipp:857-880).  pix: [ncams, N] or [N].
    want_count=False skips the host read-back of the count (returns None for it); as_bool=False returns the flags as the
    uint8 array the library wrote (what set_skipped / the exchange take: no conversion kernels in a per-step call)."""
    pix = _dev(pix, torch.int32)
    if pix.dim() == 1:
        pix = pix[None]
    ncams, n = pix.shape
    sk = torch.empty(n, dtype=torch.uint8, device="cuda")
    cnt = C.c_uint64(0)
    check(lib().upsp_projection_skipped(ncams, n, _ptr(pix), _ptr(sk), C.byref(cnt) if want_count else None,
                                        _stream()))
    return (sk.bool() if as_bool else sk), (int(cnt.value) if want_count else None)


def fix_hot_pixels(frames, thresh=4064, min_change=512, max_hot=5):
    """upsp::fix_hot_pixels in place on u16 frames [F,H,W]; returns status int32[F]."""
    assert frames.is_cuda and frames.dtype == torch.uint16 and frames.is_contiguous()
    if frames.dim() == 2:
        frames = frames[None]
    f, h, w = frames.shape
    st = torch.empty(f, dtype=torch.int32, device="cuda")
    check(lib().upsp_fix_hot_pixels(_ptr(frames), f, h, w, thresh, min_change, max_hot, _ptr(st),
                                    _stream()))
    return st


def project_frame(img, pix, weight=None):
    """upsp::project_frame (projection.ipp:883-908) for one u16 / f32 image."""
    assert img.is_cuda and img.is_contiguous()
    n = pix.numel()
    out = torch.empty(n, dtype=torch.float32, device="cuda")
    fn = lib().upsp_project_frame_u16 if img.dtype == torch.uint16 else lib().upsp_project_frame_f32
    if img.dtype not in (torch.uint16, torch.float32):
        raise TypeError("image must be uint16 or float32")
    check(fn(_ptr(img), _ptr(pix), _ptr(weight), n, _ptr(out), _stream()))
    return out


def transpose(src, out=None, ld=None, col0=0):
    """local_transpose (psp_process.cpp:647-689): [Y,X] f32 -> [X,Y]."""
    assert src.is_cuda and src.dtype == torch.float32 and src.is_contiguous()
    y, x = src.shape
    if out is None:
        out = torch.empty((x, y), dtype=torch.float32, device="cuda")
        ld = y
    check(lib().upsp_transpose_f32(_ptr(src), x, y, C.c_void_p(out.data_ptr() + 4 * col0), ld,
                                   _stream()))
    return out


def apportion(value, nbins):
    """apportion (psp_process.cpp:611-624) -> (start[], extent[])."""
    st = (C.c_int * nbins)()
    ex = (C.c_int * nbins)()
    check(lib().upsp_apportion(int(value), int(nbins), st, ex))
    return list(st), list(ex)


def series_ld(nframes, whole_rows=False):
    """Row pitch (in floats) for a node-major time-series buffer [N, nframes]: rows start on
    256-byte boundaries.  Callers that fill the rows in column chunks (process(..., col0=...) per
    64..256 frames, or any schedule other than the streamed one) get an odd multiple of 256 B, so
    that the same piece of consecutive rows does not map to the same HBM channels (a 4-KiB pitch
    measured 8 % slower than 4352 B for 256-B pieces on MI355X).  whole_rows=True: one process()
    call writes every row piece whole (streamed schedule, <= 1024 frames per call): the plain
    multiple of 256 B is best there (tools/probe/store_shapes.hip: 5.7-5.9 TB/s at 4096 B against
    5.0-5.2 at 4352 B).  Pass `torch.empty((N, ld))[:, :nframes]` as rows_t."""
    ld = (int(nframes) + 63) // 64 * 64
    if whole_rows:
        return ld
    return ld + 64 if (ld // 64) % 2 == 0 else ld


class FramePipeline:
    """Body of the psp_process phase-1 frame loop (psp_process.cpp:1743-1851)."""

    def __init__(self, ncams, width, height, nnodes, **opts):
        o = _capi.PipelineOpts()
        lib().upsp_pipeline_default_opts(C.byref(o))
        for k, v in opts.items():
            if not hasattr(o, k):
                raise TypeError("unknown pipeline option %r" % k)
            setattr(o, k, v)
        h = C.c_void_p()
        check(lib().upsp_pipeline_create(ncams, width, height, nnodes, C.byref(o), C.byref(h)))
        self._h, self.ncams, self.width, self.height, self.nnodes = h, ncams, width, height, nnodes
        self._destroy = lib().upsp_pipeline_destroy
        self.opts = o

    def close(self):
        if getattr(self, "_h", None):
            self._destroy(self._h)
            self._h = None

    __del__ = close

    def set_projection(self, cam, pix, weight=None):
        pix = _dev(pix, torch.int32)
        w = None if weight is None else _dev(weight, torch.float32)
        assert pix.numel() == self.nnodes
        # copies ordered on the current stream (no host synchronisation: the projection build and
        # the frame loop are queued back to back)
        check(lib().upsp_pipeline_set_projection_async(self._h, cam, _ptr(pix), _ptr(w), _stream()))

    def fix_hot_pixels(self, frames):
        """Queue the hot-pixel scan + repair of resident u16 frames [F,H,W] (one camera) on the
        current stream, ahead of process(..., hot_fixed=True).  It does not depend on the
        projection, so it can run on a side stream while the projection is being built."""
        assert frames.is_cuda and frames.dtype == torch.uint16 and frames.is_contiguous()
        assert tuple(frames.shape[1:]) == (self.height, self.width)
        check(lib().upsp_pipeline_fix_hot_pixels(self._h, _ptr(frames), frames.shape[0], _stream()))

    def set_active_hint(self, pix_candidates):
        """Active-pixel map from a candidate set (engine.candidate_pixels) instead of a projection: pass A
        (prescan) can then run before / while the projection is built.  None clears it."""
        self._hint = None if pix_candidates is None else _dev(pix_candidates, torch.int32)
        check(lib().upsp_pipeline_set_active_hint(self._h, _ptr(self._hint), _stream()))

    def prescan(self, frames):
        """Pass A (hot-pixel count + compact pixel series) of u16 frames [F <= 1024, H, W] on the current
        stream; the next process() call on the same tensor runs pass B + the hot-pixel fix-up only."""
        assert frames.is_cuda and frames.dtype == torch.uint16 and frames.is_contiguous()
        assert tuple(frames.shape[1:]) == (self.height, self.width)
        check(lib().upsp_pipeline_prescan(self._h, _ptr(frames), frames.shape[0], _stream()))

    def projection_target(self, cam=0):
        """int32 [N] tensor over a pipeline-owned buffer no queued launch reads (upsp_pipeline_projection_target): build the
        next projection into it (build_projection(pix_out=...)) and set_projection() takes it over without a copy.  Valid
        until that set_projection() call."""
        q = C.c_void_p()
        check(lib().upsp_pipeline_projection_target(self._h, cam, C.byref(q)))
        return torch.as_tensor(_DevArray(q.value, self.nnodes, "<i4", self), device="cuda")

    def current_projection(self, cam=0):
        """int32 [N] tensor over the projection the pipeline currently holds for camera `cam` (upsp_pipeline_projection) -- after
        step(): the one that step built.  Clone it to keep it: the buffer is reused two projections later."""
        q = C.c_void_p()
        check(lib().upsp_pipeline_projection(self._h, cam, C.byref(q)))
        return torch.as_tensor(_DevArray(q.value, self.nnodes, "<i4", self), device="cuda")

    def prepare_rows(self):
        """What pass B needs from a new projection (every node's row in the compact series, the skipped flags), queued on the
        current stream now instead of inside the next process() call (upsp_pipeline_prepare_rows): behind set_projection() on
        the stream that built the projection it runs beside pass A."""
        check(lib().upsp_pipeline_prepare_rows(self._h, _stream()))

    def row_tables(self):
        """dict(node_k int32 [N], skipped uint8 [N]): the tables prepare_rows() derived, as tensors over the pipeline's buffers
        (upsp_pipeline_row_tables; valid until the next projection / map change)."""
        nk, sk = C.c_void_p(), C.c_void_p()
        check(lib().upsp_pipeline_row_tables(self._h, C.byref(nk), C.byref(sk)))
        return dict(node_k=torch.as_tensor(_DevArray(nk.value, self.nnodes, "<i4", self), device="cuda"),
                    skipped=torch.as_tensor(_DevArray(sk.value, self.nnodes, "|u1", self), device="cuda"))

    def pixel_series(self, frames):
        """Pass A alone (upsp_pipeline_pixel_series): the REPAIRED u16 series of every active pixel over these (<= 1024)
        frames, left in the pipeline's compact buffer; the frames are repaired in place.  Returns dict(ptr = device address
        of the buffer [active pixel][cpitch], cpitch, node_k = int32 [N] tensor aliasing the pipeline's node -> row table,
        nactive = uint32 [1] tensor (rows in use)).  Valid until the next call on this pipeline."""
        comp, nk, na, cp = C.c_void_p(), C.c_void_p(), C.c_void_p(), C.c_uint32()
        if frames is None or frames.shape[0] == 0:      # only the node -> row table of the current projection
            check(lib().upsp_pipeline_pixel_series(self._h, None, 0, _stream(), C.byref(comp), C.byref(cp), C.byref(nk), C.byref(na)))
        else:
            assert frames.is_cuda and frames.dtype == torch.uint16 and frames.is_contiguous()
            assert tuple(frames.shape[1:]) == (self.height, self.width)
            check(lib().upsp_pipeline_pixel_series(self._h, _ptr(frames), frames.shape[0], _stream(), C.byref(comp), C.byref(cp),
                                                   C.byref(nk), C.byref(na)))
        return dict(ptr=comp.value, cpitch=int(cp.value), rows=min(self.nnodes, self.width * self.height), owner=self,
                    node_k=torch.as_tensor(_DevArray(nk.value, self.nnodes, "<i4", self), device="cuda"),
                    nactive=torch.as_tensor(_DevArray(na.value, 1, "<u4", self), device="cuda"))

    def series_frames_max(self):
        """Largest frame count one pixel_series() / prescan() call takes (upsp_pipeline_series_frames_max)."""
        n = lib().upsp_pipeline_series_frames_max(self._h)
        check(min(n, 0))
        return int(n)

    def set_skipped(self, skipped):
        sk = None if skipped is None else _dev(skipped, torch.uint8)
        check(lib().upsp_pipeline_set_skipped(self._h, _ptr(sk)))

    def set_row_map(self, rowmap):
        """Packed time series: rowmap int32 [N] = row of rows_t per node, < 0 = not stored."""
        self._rowmap = None if rowmap is None else _dev(rowmap, torch.int32)
        check(lib().upsp_pipeline_set_row_map_async(self._h, _ptr(self._rowmap), _stream()))

    def set_row_padding(self, on=True):
        """The series buffers passed to process() were allocated as `torch.empty((N, series_ld(F)))[:, :F]`: the columns
        between the last frame of a call and the next 128-byte boundary of the row are padding the row pass may write
        (upsp_pipeline_set_row_padding: whole 128-byte lines at the end of every row)."""
        check(lib().upsp_pipeline_set_row_padding(self._h, 1 if on else 0))

    def set_scan_split(self, on=True):
        """Pass A in two launches (upsp_pipeline_set_scan_split): for a frame loop that runs beside other kernels, e.g. a projection
        build on a stream of its own; slower when pass A has the device to itself."""
        check(lib().upsp_pipeline_set_scan_split(self._h, 1 if on else 0))

    def set_overlap_source(self, src):
        """P3D adjust_solution: src int32 [N] (grids.P3DModel.overlap_source()); None = off."""
        self._src = None if src is None else _dev(src, torch.int32)
        check(lib().upsp_pipeline_set_overlap_source(self._h, _ptr(self._src)))

    def set_reference(self, cam, ref32f):
        ref = _dev(ref32f, torch.float32)
        assert ref.numel() == self.width * self.height
        check(lib().upsp_pipeline_set_reference(self._h, cam, _ptr(ref)))

    def set_patches(self, cam, clusters):
        """clusters: list of dict(bx,by,ix,iy) integer pixel lists (PatchClusters members)."""
        b_off, i_off = [0], [0]
        bx, by, ix, iy = [], [], [], []
        for cl in clusters:
            bx += list(cl["bx"]); by += list(cl["by"]); ix += list(cl["ix"]); iy += list(cl["iy"])
            b_off.append(len(bx)); i_off.append(len(ix))
        arr = [np.asarray(a, dtype=np.int32) for a in (b_off, bx, by, i_off, ix, iy)]
        p = [a.ctypes.data_as(C.c_void_p) for a in arr]
        check(lib().upsp_pipeline_set_patches(self._h, cam, len(clusters), *p))

    def process(self, frames, first_frame=0, rows=None, rows_t=None, col0=0, want_rows=True,
                warps=None, hot_fixed=False, ecc_iters=None):
        """frames: list (one per camera) of u16 tensors [F,H,W] (modified in place by the
        hot-pixel fix, like the reference).  Returns rows [F,N] f32 (or None).
        hot_fixed: the frames already went through fix_hot_pixels() -- skip the scan.
        warps / ecc_iters: optional outputs of the registration stage, f32 [F, ncams, 6] and int32 [F, ncams]
        (warp matrix and iteration count of cv::findTransformECC per frame)."""
        if hot_fixed and self.opts.hot_enable:
            check(lib().upsp_pipeline_set_hot_enable(self._h, 0))
            try:
                return self.process(frames, first_frame, rows, rows_t, col0, want_rows, warps, ecc_iters=ecc_iters)
            finally:
                check(lib().upsp_pipeline_set_hot_enable(self._h, 1))
        if ecc_iters is not None:
            assert ecc_iters.is_cuda and ecc_iters.dtype == torch.int32 and ecc_iters.is_contiguous()
            check(lib().upsp_pipeline_set_ecc_iterations_out(self._h, _ptr(ecc_iters)))
            try:
                return self.process(frames, first_frame, rows, rows_t, col0, want_rows, warps)
            finally:
                check(lib().upsp_pipeline_set_ecc_iterations_out(self._h, None))
        if isinstance(frames, torch.Tensor):
            frames = [frames]
        assert len(frames) == self.ncams
        f = frames[0].shape[0]
        for fr in frames:
            assert fr.is_cuda and fr.dtype == torch.uint16 and fr.is_contiguous()
            assert tuple(fr.shape) == (f, self.height, self.width)
        ptrs = (C.c_void_p * self.ncams)(*[fr.data_ptr() for fr in frames])
        if rows_t is not None and rows_t.dtype == torch.uint16:
            # wire format of the time-series exchange (upsp_pipeline_process_u16): the library
            # refuses it unless every stored value is an exact 16-bit integer
            assert rows_t.is_cuda and (rows_t.shape[1] <= 1 or rows_t.stride(1) == 1)
            check(lib().upsp_pipeline_process_u16(self._h, ptrs, f, int(first_frame), _ptr(rows_t),
                                                  rows_t.stride(0), col0, _ptr(warps), _stream()))
            return None
        if rows is None and want_rows:
            rows = torch.empty((f, self.nnodes), dtype=torch.float32, device="cuda")
        ld = 0 if rows_t is None else rows_t.stride(0)
        check(lib().upsp_pipeline_process(self._h, ptrs, f, int(first_frame), _ptr(rows),
                                          _ptr(rows_t), ld, col0, _ptr(warps), _stream()))
        return rows

    def step(self, bvh, cam, nodes, normals, tri_nodes, frames, rows_t=None, first_frame=0, col0=0, oblique_angle=70.0,
             datanode=None, finals=None, nframes_total=0, frames_hook=None, tail_hook=None):
        """One step of a frame loop whose projection is rebuilt per batch (upsp_pipeline_step, include/upsp_gpu.h): candidate
        pixels, map, projection build and hand-over on the pipeline's own high-priority stream, pass A + repair + pass B on the
        current stream, every ordering event inside the library.  nodes / normals / tri_nodes: device tensors (f32 [N,3],
        f32 [N,3], int32 [3T]); frames: u16 [F <= 1024, H, W] (repaired in place); rows_t: f32 [N, >= F] node-major (None: no
        pass B, the series stay in the compact buffer for pixel_series()).  finals: (avg, rms) f32 [N] tensors written with the
        next step or by step_finish().  frames_hook / tail_hook: callables taking a torch stream (the side stream), see the
        header."""
        assert frames.is_cuda and frames.dtype == torch.uint16 and frames.is_contiguous()
        assert tuple(frames.shape[1:]) == (self.height, self.width)
        for t_, dt in ((nodes, torch.float32), (normals, torch.float32), (tri_nodes, torch.int32)):
            assert t_.is_cuda and t_.dtype == dt and t_.is_contiguous()
        a = _capi.StepArgs()
        a.bvh = bvh.handle
        a.cam = C.pointer(cam)
        a.d_nodes, a.d_normals, a.d_tri_nodes = nodes.data_ptr(), normals.data_ptr(), tri_nodes.data_ptr()
        a.d_datanode = None if datanode is None else datanode.data_ptr()
        a.oblique_thresh = oblique_threshold(oblique_angle)
        a.nframes = frames.shape[0]
        a.d_frames = frames.data_ptr()
        a.first_frame = int(first_frame)
        if rows_t is not None:
            assert rows_t.is_cuda and rows_t.dtype == torch.float32 and (rows_t.shape[1] <= 1 or rows_t.stride(1) == 1)
            a.d_rows_t, a.ld_t, a.col0 = rows_t.data_ptr(), rows_t.stride(0), int(col0)
        if finals is not None:
            a.d_avg, a.d_rms = finals[0].data_ptr(), finals[1].data_ptr()
        a.nframes_total = int(nframes_total)

        def wrap(fn):
            def call(_user, stream_ptr):
                st = torch.cuda.ExternalStream(stream_ptr)
                with torch.cuda.stream(st):
                    fn(st)
            return _capi.StepHook(call)
        hooks = [wrap(frames_hook) if frames_hook else _capi.StepHook(), wrap(tail_hook) if tail_hook else _capi.StepHook()]
        a.frames_hook, a.tail_hook = hooks
        check(lib().upsp_pipeline_step(self._h, C.byref(a), _stream()))
        self._step_keep = (finals, rows_t, frames)      # (buffers the queued launches write)

    def step_mark_end(self):
        """The end of a step issued with rows_t=None, on the current stream (upsp_pipeline_step_mark_end)."""
        check(lib().upsp_pipeline_step_mark_end(self._h, _stream()))

    def step_finish(self):
        """The finals the last step() left, and the current stream ordered behind the pipeline's side stream."""
        check(lib().upsp_pipeline_step_finish(self._h, _stream()))

    def ecc_stats(self):
        """Registration statistics since creation: dict(frame_iterations, frames)."""
        a, b = C.c_uint64(), C.c_uint64()
        check(lib().upsp_pipeline_ecc_stats(self._h, C.byref(a), C.byref(b)))
        return dict(frame_iterations=a.value, frames=b.value)

    def accumulators(self):
        """(sum, sumsq) as float64 tensors aliasing the pipeline's device buffers."""
        a, b = C.c_void_p(), C.c_void_p()
        check(lib().upsp_pipeline_accumulators_async(self._h, C.byref(a), C.byref(b), _stream()))
        return _alias_f64(a.value, self.nnodes, self), _alias_f64(b.value, self.nnodes, self)

    def reset(self, deferred=False):
        """Zero the accumulators.  deferred=True: no device work and no wait now -- the next call that uses them clears (or, in the
        one-camera streamed loop, writes) them on its own stream; tensors from an earlier accumulators() call must not be read
        or reduced until accumulators() has been called again (upsp_pipeline_reset_deferred, include/upsp_gpu.h)."""
        check((lib().upsp_pipeline_reset_deferred if deferred else lib().upsp_pipeline_reset)(self._h))

    def finalize(self, nframes_total):
        avg = torch.empty(self.nnodes, dtype=torch.float32, device="cuda")
        rms = torch.empty(self.nnodes, dtype=torch.float32, device="cuda")
        check(lib().upsp_pipeline_finalize(self._h, int(nframes_total), _ptr(avg), _ptr(rms),
                                           _stream()))
        return avg, rms


class _DevArray:
    """__cuda_array_interface__ view of a raw device pointer (keeps `owner` alive)."""

    def __init__(self, ptr, n, typestr, owner):
        self.__cuda_array_interface__ = dict(shape=(n,), typestr=typestr, data=(ptr, False),
                                             version=2)
        self._owner = owner


def _alias_f64(ptr, n, owner):
    t = torch.as_tensor(_DevArray(ptr, n, "<f8", owner), device="cuda")
    t._upsp_owner = owner
    return t


def register_pixel(ref32f, frame_u16, max_iters=50, eps=1e-3, interp=1):
    """upsp::register_pixel (cpp/lib/registration.cpp:32-81) for one frame.
    Returns (registered u16 frame, warp 2x3 float32 numpy, iterations)."""
    assert ref32f.is_cuda and ref32f.dtype == torch.float32 and ref32f.is_contiguous()
    assert frame_u16.is_cuda and frame_u16.dtype == torch.uint16 and frame_u16.is_contiguous()
    h, w = frame_u16.shape
    out = torch.empty_like(frame_u16)
    warp = (C.c_float * 6)()
    rc = lib().upsp_register_pixel_u16(_ptr(ref32f), _ptr(frame_u16), h, w, int(max_iters),
                                       float(eps), int(interp), _ptr(out), warp, _stream())
    if rc < 0:
        check(rc)
    return out, np.array(list(warp), dtype=np.float32).reshape(2, 3), rc


def blur_u16(frames_u16, k=5):
    """convertTo(CV_32F) + cv::GaussianBlur(Size(k,k), 0) of u16 frames [F,H,W] (or [H,W]) in one pass: f32 tensor of
    the same shape (upsp_blur_u16; k = 5 is the registration's pre-blur kernel, cpp/lib/registration.cpp:57-60)."""
    assert frames_u16.is_cuda and frames_u16.dtype == torch.uint16 and frames_u16.is_contiguous()
    fr = frames_u16 if frames_u16.dim() == 3 else frames_u16[None]
    f, h, w = fr.shape
    out = torch.empty((f, h, w), dtype=torch.float32, device="cuda")
    check(lib().upsp_blur_u16(_ptr(fr), _ptr(out), f, h, w, int(k), _stream()))
    return out if frames_u16.dim() == 3 else out[0]


def blur(img32f, k, box=False):
    """cv::GaussianBlur(img,img,Size(k,k),0) / cv::blur(img,img,Size(k,k)) (psp_process.cpp:1802-1807)."""
    assert img32f.is_cuda and img32f.dtype == torch.float32 and img32f.is_contiguous()
    h, w = img32f.shape
    out = torch.empty_like(img32f)
    check(lib().upsp_blur_f32(_ptr(img32f), _ptr(out), h, w, int(k), int(bool(box)), _stream()))
    return out


def _cluster_arrays(clusters):
    b_off, i_off = [0], [0]
    bx, by, ix, iy = [], [], [], []
    for cl in clusters:
        bx += list(cl["bx"]); by += list(cl["by"]); ix += list(cl["ix"]); iy += list(cl["iy"])
        b_off.append(len(bx)); i_off.append(len(ix))
    return [np.asarray(a, dtype=np.int32) for a in (b_off, bx, by, i_off, ix, iy)]


def patch(img32f, clusters):
    """PatchClusters<float>::operator() (cpp/lib/patches.ipp:98-165), in place on a f32 image [H, W] or on a stack of
    images [F, H, W]."""
    assert img32f.is_cuda and img32f.dtype == torch.float32 and img32f.is_contiguous() and img32f.dim() in (2, 3)
    h, w = img32f.shape[-2:]
    arr = _cluster_arrays(clusters)
    check(lib().upsp_patch_frames_f32(_ptr(img32f), 1 if img32f.dim() == 2 else img32f.shape[0], h, w, len(clusters),
                                      *[a.ctypes.data_as(C.c_void_p) for a in arr], _stream()))
    return img32f


# ----------------------------------------------------------------------------- phase 2 --
class TransPolyFitter:
    """upsp::TransPolyFitter<float> on the GPU (cpp/include/filtering.h:24-87,
    cpp/lib/filtering.ipp:12-79): least-squares polynomial of `degree` over x = f/n_frames
    through every row of node-major data."""

    def __init__(self, n_frames, degree, n_pts):
        if not 0 <= int(degree) <= 7:
            raise ValueError("polynomial degree must be 0..7")
        self.n_frames, self.degree, self.n_pts = int(n_frames), int(degree), int(n_pts)
        self.poly = torch.zeros((self.n_pts, self.degree + 1), dtype=torch.float32, device="cuda")

    def eval_fit(self, data, n_pts=None, curr_pt=0):
        """data: [n_pts, n_frames] f32 (node-major chunk).  Returns the fit values
        [n_pts, n_frames] and stores the coefficients of points curr_pt.. (filtering.ipp:48-79)."""
        data = _dev(data, torch.float32).reshape(-1, self.n_frames)
        n = data.shape[0] if n_pts is None else int(n_pts)
        out = torch.empty((n, self.n_frames), dtype=torch.float32, device="cuda")
        poly = torch.empty((n, self.degree + 1), dtype=torch.float32, device="cuda")
        check(lib().upsp_transpoly_fit(_ptr(data), self.n_frames, n, self.n_frames, self.degree,
                                       _ptr(out), self.n_frames, _ptr(poly), _stream()))
        self.poly[curr_pt:curr_pt + n] = poly
        return out

    def skip_fit(self, curr_pt):
        self.poly[curr_pt] = 0.0


def phase2_pressure(intensity_t, iref, coverage, paint_cal, qbar, ps, steady=None, model_temp=70.0,
                    degree=6, out=None, want=("avg", "rms", "gain")):
    """Phase-2 node loop (cpp/exec/psp_process.cpp:2452-2507) over a node-major slice.

    intensity_t [n, F] (may be a column slice of a wider buffer: stride(0) is honoured);
    iref/coverage/steady [n]; model_temp: scalar or [n].  out: destination (defaults to a new
    tensor; pass intensity_t itself to run in place).  Returns dict(pressure_t, sum, sumsq, ...)."""
    if intensity_t.dim() != 2 or intensity_t.stride(1) != 1 or intensity_t.dtype != torch.float32:
        raise ValueError("intensity_t must be a 2-D f32 tensor with unit column stride")
    n, F = intensity_t.shape
    if out is None:
        out = torch.empty((n, F), dtype=torch.float32, device="cuda")
    if out.shape != (n, F) or out.stride(1) != 1 or out.dtype != torch.float32:
        raise ValueError("bad output tensor")
    iref = _dev(iref, torch.float32)
    coverage = _dev(coverage, torch.float32)
    steady_t = None if steady is None else _dev(steady, torch.float32)
    temp_t, temp_s = None, 0.0
    if isinstance(model_temp, (int, float)):
        temp_s = float(model_temp)
    else:
        temp_t = _dev(model_temp, torch.float32)
    for t in (iref, coverage, steady_t, temp_t):
        if t is not None and t.numel() != n:
            raise ValueError("per-node input has the wrong length")
    cal = (C.c_float * 6)(*[float(v) for v in paint_cal])
    res = dict(pressure_t=out,
               sum=torch.empty(n, dtype=torch.float64, device="cuda"),
               sumsq=torch.empty(n, dtype=torch.float64, device="cuda"))
    for k in want:
        res[k] = torch.empty(n, dtype=torch.float32, device="cuda")
    check(lib().upsp_phase2_pressure(
        _ptr(intensity_t), intensity_t.stride(0), n, F, _ptr(iref), _ptr(coverage), _ptr(steady_t),
        _ptr(temp_t), temp_s, cal, float(qbar), float(ps), int(degree), _ptr(out), out.stride(0),
        _ptr(res["sum"]), _ptr(res["sumsq"]), _ptr(res.get("avg")), _ptr(res.get("rms")),
        _ptr(res.get("gain")), _stream()))
    return res


# ------------------------------------------------------------------------ nearest node --
def nearest_nodes(nodes, queries, want_dist=False):
    """kd_nearest over all model nodes (cpp/raycast/pspKdtree.c:284-372) as an exhaustive
    GPU scan.  nodes: [N,3] f32 device tensor (or array); queries: [Q,3] (double).
    Returns int32 [Q] (and the squared distances, double [Q], if want_dist)."""
    nodes = _dev(nodes, torch.float32).reshape(-1, 3)
    q = _dev(queries, torch.float64).reshape(-1, 3)
    idx = torch.empty(q.shape[0], dtype=torch.int32, device="cuda")
    d2 = torch.empty(q.shape[0], dtype=torch.float64, device="cuda") if want_dist else None
    check(lib().upsp_nearest_nodes(_ptr(nodes), nodes.shape[0], _ptr(q), q.shape[0], _ptr(idx),
                                   _ptr(d2), _stream()))
    return (idx, d2) if want_dist else idx


def interpolate_idw(src_nodes, src_data, query_nodes, k=10, p=2.0, want_neighbors=False):
    """upsp::interpolate (cpp/lib/interpolation.ipp:16-70): k-nearest inverse-distance weighting.
    src_nodes [M,3] / src_data [M]: host arrays; query_nodes [Q,3]: array or device tensor.
    Returns f32 [Q] on the device (and the neighbour ids int32 [Q,k] if requested)."""
    src_nodes = np.ascontiguousarray(src_nodes, dtype=np.float32).reshape(-1, 3)
    src_data = np.ascontiguousarray(src_data, dtype=np.float32).reshape(-1)
    if src_data.size != src_nodes.shape[0]:
        raise ValueError("one value per source node")
    q = _dev(query_nodes, torch.float32).reshape(-1, 3)
    out = torch.empty(q.shape[0], dtype=torch.float32, device="cuda")
    nb = torch.empty((q.shape[0], int(k)), dtype=torch.int32, device="cuda") if want_neighbors else None
    check(lib().upsp_interpolate_idw(src_nodes.ctypes.data_as(C.c_void_p), src_data.ctypes.data_as(C.c_void_p),
                                     src_nodes.shape[0], _ptr(q), q.shape[0], int(k), C.c_float(p), _ptr(out),
                                     _ptr(nb), _stream()))
    return (out, nb) if want_neighbors else out
