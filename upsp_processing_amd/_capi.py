"""ctypes view of libupsp_gpu.so -- the C ABI declared in include/upsp_gpu.h.

There is deliberately NO fallback: if the HIP library is missing or fails to load,
importing the compute path raises.  (The CPU oracle under oracle/ is test
infrastructure and is never imported from here.)
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libupsp_gpu.so")

UPSP_OK = 0
STATUS = {0: "UPSP_OK", -1: "UPSP_ERR_INVALID", -2: "UPSP_ERR_EMPTY", -3: "UPSP_ERR_DEPTH",
          -4: "UPSP_ERR_HIP", -5: "UPSP_ERR_NO_DEVICE", -6: "UPSP_ERR_DIVERGED", -7: "UPSP_ERR_INTERNAL"}


class UpspError(RuntimeError):
    def __init__(self, status, msg):
        super().__init__("%s: %s" % (STATUS.get(status, status), msg))
        self.status = status


class BvhInfo(C.Structure):
    _fields_ = [("ntris", C.c_uint64), ("n_ref_nodes", C.c_uint32), ("n_gpu_nodes", C.c_uint32),
                ("depth", C.c_uint32), ("max_leaf", C.c_uint32),
                ("bounds_min", C.c_float * 3), ("bounds_max", C.c_float * 3),
                ("device_bytes", C.c_uint64), ("build_seconds", C.c_double)]


class Hits(C.Structure):
    _fields_ = [("hit", C.c_void_p), ("t", C.c_void_p), ("prim", C.c_void_p),
                ("uvw", C.c_void_p), ("pos", C.c_void_p), ("nrm", C.c_void_p)]


class Camera(C.Structure):
    _fields_ = [("K", C.c_double * 9), ("dist", C.c_double * 5), ("R", C.c_double * 9),
                ("t", C.c_double * 3), ("width", C.c_int32), ("height", C.c_int32)]


class PipelineOpts(C.Structure):
    _fields_ = [("hot_enable", C.c_int32), ("hot_thresh", C.c_int32),
                ("hot_min_change", C.c_int32), ("hot_max", C.c_int32),
                ("registration", C.c_int32), ("ecc_max_iters", C.c_int32),
                ("ecc_eps", C.c_double), ("interp", C.c_int32),
                ("filter", C.c_int32), ("filter_size", C.c_int32), ("patch", C.c_int32),
                ("fused_scan", C.c_int32), ("compact_mb", C.c_int32), ("reserved", C.c_int32 * 3)]


StepHook = C.CFUNCTYPE(None, C.c_void_p, C.c_void_p)      # upsp_step_hook(user, stream)


class StepArgs(C.Structure):                               # upsp_step_args
    _fields_ = [("bvh", C.c_void_p), ("cam", C.POINTER(Camera)), ("d_nodes", C.c_void_p), ("d_normals", C.c_void_p),
                ("d_datanode", C.c_void_p), ("d_tri_nodes", C.c_void_p), ("oblique_thresh", C.c_float), ("nframes", C.c_int32),
                ("d_frames", C.c_void_p), ("first_frame", C.c_int64), ("d_rows_t", C.c_void_p), ("ld_t", C.c_int64),
                ("col0", C.c_int64), ("d_avg", C.c_void_p), ("d_rms", C.c_void_p), ("nframes_total", C.c_uint64),
                ("frames_hook", StepHook), ("frames_user", C.c_void_p), ("tail_hook", StepHook), ("tail_user", C.c_void_p)]


_vp, _sz, _i, _i64, _u64p = C.c_void_p, C.c_size_t, C.c_int, C.c_int64, C.POINTER(C.c_uint64)

# name -> (restype, argtypes); every function include/upsp_gpu.h declares
SIGNATURES = {
    "upsp_last_error": (C.c_char_p, []),
    "upsp_device_info": (_i, [C.POINTER(_i), C.c_char_p, C.POINTER(_i)]),
    "upsp_version": (_i, []),
    "upsp_bvh_create": (_i, [_vp, _sz, C.POINTER(_vp)]),
    "upsp_bvh_destroy": (None, [_vp]),
    "upsp_bvh_share": (_i, [_vp, C.POINTER(_vp)]),
    "upsp_bvh_get_info": (_i, [_vp, C.POINTER(BvhInfo)]),
    "upsp_bvh_intersect": (_i, [_vp, _vp, _i, _vp, _sz, C.POINTER(Hits), _vp]),
    "upsp_bvh_intersect_host": (_i, [_vp, _vp, _i, _vp, _sz, C.POINTER(Hits)]),
    "upsp_bvh_occluded": (_i, [_vp, _vp, _i, _vp, _sz, _vp, _vp]),
    "upsp_bvh_occluded_host": (_i, [_vp, _vp, _i, _vp, _sz, _vp]),
    "upsp_bvh_enable_stats": (_i, [_vp, _i]),
    "upsp_bvh_check": (_i, [_vp, _vp]),
    "upsp_bvh_last_filter_stats": (_i, [_vp, _u64p, _u64p]),
    "upsp_bvh_last_stats": (_i, [_vp, _u64p, _u64p, _u64p]),
    "upsp_projection_build": (_i, [_vp, C.POINTER(Camera), _vp, _vp, _vp, _vp, _sz, C.c_float,
                                   _vp, _vp, _vp, _u64p, _vp]),
    "upsp_projection_fetch_counts": (_i, [_vp, _vp, _vp, _vp, _vp]),
    "upsp_projection_last_counts": (_i, [_vp, _u64p, _u64p]),
    "upsp_projection_weights": (_i, [_i, _sz, _vp, _vp, _vp, _vp, _vp, _i, _vp]),
    "upsp_projection_skipped": (_i, [_i, _sz, _vp, _vp, _u64p, _vp]),
    "upsp_camera_center": (_i, [C.POINTER(Camera), _vp]),
    "upsp_project_points_host": (_i, [C.POINTER(Camera), _vp, _sz, _vp]),
    "upsp_pipeline_default_opts": (None, [C.POINTER(PipelineOpts)]),
    "upsp_pipeline_create": (_i, [_i, _i, _i, _sz, C.POINTER(PipelineOpts), C.POINTER(_vp)]),
    "upsp_pipeline_destroy": (None, [_vp]),
    "upsp_pipeline_set_projection": (_i, [_vp, _i, _vp, _vp]),
    "upsp_pipeline_set_projection_async": (_i, [_vp, _i, _vp, _vp, _vp]),
    "upsp_pipeline_fix_hot_pixels": (_i, [_vp, _vp, _i, _vp]),
    "upsp_pipeline_set_hot_enable": (_i, [_vp, _i]),
    "upsp_fill_rows_f32": (_i, [C.c_float, _sz, _i, _vp, _vp, C.c_longlong, _vp]),
    "upsp_scatter_rows_f32": (_i, [_vp, _sz, _i, _vp, _vp, C.c_longlong, _vp]),
    "upsp_scatter_rows_u16": (_i, [_vp, _sz, _i, _vp, _vp, C.c_longlong, _vp]),
    "upsp_pipeline_set_row_map": (_i, [_vp, _vp]),
    "upsp_pipeline_set_row_map_async": (_i, [_vp, _vp, _vp]),
    "upsp_pipeline_set_row_padding": (_i, [_vp, _i]),
    "upsp_pipeline_set_overlap_source": (_i, [_vp, _vp]),
    "upsp_pipeline_set_skipped": (_i, [_vp, _vp]),
    "upsp_pipeline_set_reference": (_i, [_vp, _i, _vp]),
    "upsp_pipeline_set_patches": (_i, [_vp, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp]),
    "upsp_pipeline_process": (_i, [_vp, C.POINTER(_vp), _i, _i64, _vp, _vp, _i64, _i64, _vp, _vp]),
    "upsp_pipeline_ecc_stats": (_i, [_vp, _vp, _vp]),
    "upsp_pipeline_set_ecc_iterations_out": (_i, [_vp, _vp]),
    "upsp_pipeline_set_active_hint": (_i, [_vp, _vp, _vp]),
    "upsp_pipeline_prescan": (_i, [_vp, _vp, _i, _vp]),
    "upsp_pipeline_prepare_rows": (_i, [_vp, _vp]),
    "upsp_pipeline_set_scan_split": (_i, [_vp, _i]),
    "upsp_pipeline_row_tables": (_i, [_vp, _vp, _vp]),
    "upsp_pipeline_projection_target": (_i, [_vp, _i, _vp]),
    "upsp_projection_candidate_pixels": (_i, [C.POINTER(Camera), _vp, _vp, _sz, _vp, _vp]),
    "upsp_projection_candidate_pixels_oblique": (_i, [C.POINTER(Camera), _vp, _vp, _vp, _sz, C.c_float, _vp, _vp]),
    "upsp_feed_create": (_i, [_sz, _i, C.POINTER(_vp)]),
    "upsp_feed_destroy": (None, [_vp]),
    "upsp_feed_acquire": (_i, [_vp, C.POINTER(_i), C.POINTER(_vp)]),
    "upsp_feed_commit": (_i, [_vp, _i, _sz, _vp, C.POINTER(_vp)]),
    "upsp_feed_abort": (_i, [_vp, _i]),
    "upsp_feed_release": (_i, [_vp, _i, _vp]),
    "upsp_pipeline_process_u16": (_i, [_vp, C.POINTER(_vp), _i, _i64, _vp, _i64, _i64, _vp, _vp]),
    "upsp_pipeline_accumulators": (_i, [_vp, C.POINTER(_vp), C.POINTER(_vp)]),
    "upsp_pipeline_accumulators_async": (_i, [_vp, C.POINTER(_vp), C.POINTER(_vp), _vp]),
    "upsp_pipeline_projection": (_i, [_vp, _i, C.POINTER(_vp)]),
    "upsp_pipeline_reset": (_i, [_vp]),
    "upsp_pipeline_step": (_i, [_vp, C.POINTER(StepArgs), _vp]),
    "upsp_pipeline_step_mark_end": (_i, [_vp, _vp]),
    "upsp_pipeline_step_finish": (_i, [_vp, _vp]),
    "upsp_pipeline_reset_deferred": (_i, [_vp]),
    "upsp_pipeline_finalize": (_i, [_vp, C.c_uint64, _vp, _vp, _vp]),
    "upsp_fix_hot_pixels": (_i, [_vp, _i, _i, _i, _i, _i, _i, _vp, _vp]),
    "upsp_project_frame_u16": (_i, [_vp, _vp, _vp, _sz, _vp, _vp]),
    "upsp_project_frame_f32": (_i, [_vp, _vp, _vp, _sz, _vp, _vp]),
    "upsp_transpose_f32": (_i, [_vp, _i64, _i64, _vp, _i64, _vp]),
    "upsp_apportion": (_i, [_i, _i, _vp, _vp]),
    "upsp_register_pixel_u16": (_i, [_vp, _vp, _i, _i, _i, C.c_double, _i, _vp, _vp, _vp]),
    "upsp_blur_f32": (_i, [_vp, _vp, _i, _i, _i, _i, _vp]),
    "upsp_blur_u16": (_i, [_vp, _vp, _i, _i, _i, _i, _vp]),
    "upsp_patch_f32": (_i, [_vp, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "upsp_patch_frames_f32": (_i, [_vp, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "upsp_unpack_10bit": (_i, [_vp, _i, _sz, _vp, _vp, _vp]),
    "upsp_unpack_12bit": (_i, [_vp, _i, _sz, _vp, _i, _vp, _vp]),
    "upsp_transpoly_fit": (_i, [_vp, C.c_longlong, _sz, _i, _i, _vp, C.c_longlong, _vp, _vp]),
    "upsp_phase2_pressure": (_i, [_vp, C.c_longlong, _sz, _i, _vp, _vp, _vp, _vp, C.c_float, _vp,
                                  C.c_float, C.c_float, _i, _vp, C.c_longlong, _vp, _vp, _vp, _vp, _vp, _vp]),
    "upsp_bvh_set_tri_nodes": (_i, [_vp, _vp, _sz, _vp]),
    "upsp_interpolate_idw": (_i, [_vp, _vp, _sz, _vp, _sz, _i, C.c_float, _vp, _vp, _vp]),
    "upsp_nearest_nodes": (_i, [_vp, _sz, _vp, _sz, _vp, _vp, _vp]),
    "upsp_comm_unique_id": (_i, [_vp]),
    "upsp_comm_create": (_i, [_vp, _i, _i, C.POINTER(_vp)]),
    "upsp_comm_from_nccl": (_i, [_vp, C.POINTER(_vp)]),
    "upsp_comm_create_local": (_i, [_i, C.POINTER(_vp)]),
    "upsp_comm_destroy": (None, [_vp]),
    "upsp_comm_rank": (_i, [_vp, C.POINTER(_i), C.POINTER(_i)]),
    "upsp_allreduce_sums": (_i, [_vp, _vp, _vp, _sz, _vp]),
    "upsp_exchange_create": (_i, [_vp, C.c_int64, C.c_int64, _i, C.POINTER(_vp)]),
    "upsp_exchange_destroy": (None, [_vp]),
    "upsp_exchange_layout": (_i, [_vp, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "upsp_exchange_chunk": (_i, [_vp, _i, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "upsp_exchange_set_skipped": (_i, [_vp, _vp, _i, _vp]),
    "upsp_exchange_rows": (_i, [_vp, C.POINTER(_vp), C.POINTER(C.c_int64)]),
    "upsp_exchange_submit": (_i, [_vp, _vp, _i, _vp]),
    "upsp_exchange_finish": (_i, [_vp, _vp, C.c_int64, _vp]),
    "upsp_exchange_verify": (_i, [_vp, _vp]),
    "upsp_exchange_bytes": (_i, [_vp, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    "upsp_pipeline_pixel_series": (_i, [_vp, _vp, _i, _vp, C.POINTER(_vp), C.POINTER(C.c_uint32), C.POINTER(_vp), C.POINTER(_vp)]),
    "upsp_pipeline_series_frames_max": (_i, [_vp]),
    "upsp_rows_from_pixel_series": (_i, [_vp, C.c_uint32, _vp, _vp, _sz, C.c_int64, _vp, C.c_int64, _vp, _vp, _vp]),
    "upsp_rows_from_pixel_blocks": (_i, [_vp, _vp, _vp, _i, _vp, _vp, _sz, _vp, C.c_int64, C.c_int64, _vp, _vp, _vp]),
    "upsp_exchange_set_pixels": (_i, [_vp, _vp, _vp, _i, _vp]),
    "upsp_exchange_pixel_rows": (_i, [_vp, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "upsp_exchange_submit_pixels": (_i, [_vp, _vp, C.c_uint32, _i, _vp]),
    "upsp_exchange_finish_pixels": (_i, [_vp, _vp, C.c_int64, _vp, _vp, _vp]),
    "upsp_exchange_set_row_padding": (_i, [_vp, _i]),
    "upsp_phase_begin": (_i, [C.c_char_p]),
    "upsp_phase_end": (_i, [C.POINTER(C.c_double)]),
    "upsp_timing_enable": (_i, [_i]),
    "upsp_bandwidth_probe": (_i, [_i, _vp, _sz, _i, C.POINTER(C.c_float), _vp]),
    "upsp_copy_probe": (_i, [_vp, _vp, C.c_size_t, _i, _vp, _vp]),
    "upsp_comm_library": (_i, [C.c_char_p, _sz]),
    "upsp_timing_report": (_i, [C.c_char_p, _sz]),
}

_lib = None


def lib():
    """Load libupsp_gpu.so; raises (never falls back) when it is missing."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                "libupsp_gpu.so not found at %s -- build it with "
                "`python -m upsp_processing_amd.build` (there is no CPU fallback)" % LIB_PATH)
        L = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)  # AttributeError if the ABI lost a symbol
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def check(rc):
    if rc != UPSP_OK:
        raise UpspError(rc, (lib().upsp_last_error() or b"").decode())
    return rc


def make_camera(K, dist, R, t, width, height):
    import numpy as np
    cam = Camera()
    cam.K[:] = np.asarray(K, dtype=np.float64).ravel().tolist()
    d = np.zeros(5)
    dd = np.asarray(dist, dtype=np.float64).ravel()
    d[:min(5, dd.size)] = dd[:5]
    cam.dist[:] = d.tolist()
    cam.R[:] = np.asarray(R, dtype=np.float64).ravel().tolist()
    cam.t[:] = np.asarray(t, dtype=np.float64).ravel().tolist()
    cam.width, cam.height = int(width), int(height)
    return cam


def device_info():
    n, cus = C.c_int(0), C.c_int(0)
    arch = C.create_string_buffer(64)
    check(lib().upsp_device_info(C.byref(n), arch, C.byref(cus)))
    return dict(n_devices=n.value, arch=arch.value.decode(), n_cus=cus.value)


class phase:
    """`with _capi.phase("phase 1: frame loop"):` -- a roctx range + (UPSP_PHASE_TIMES) the reference's timedBarrierPoint
    line (cpp/exec/psp_process.cpp:585-606); the GPU is synchronised at the end so that the figure holds the phase's work."""

    def __init__(self, label, sync=True):
        self.label, self.sync, self.seconds = label, sync, None

    def __enter__(self):
        check(lib().upsp_phase_begin(self.label.encode()))
        return self

    def __exit__(self, *exc):
        if self.sync:
            import torch
            if torch.cuda.is_available():
                torch.cuda.synchronize()
        s = C.c_double()
        check(lib().upsp_phase_end(C.byref(s)))
        self.seconds = s.value
        return False


def timing_enable(on=True):
    check(lib().upsp_timing_enable(int(bool(on))))


def timing_report(spread=False):
    """dict name -> (calls, total_ms) of the kernels timed since timing_enable(True);
    spread=True: (calls, total_ms, min_ms, median_ms, max_ms) -- the spread of the single timed spans."""
    buf = C.create_string_buffer(1 << 16)
    check(lib().upsp_timing_report(buf, len(buf)))
    out = {}
    for line in buf.value.decode().splitlines():
        name, n, ms, lo, med, hi = line.rsplit(" ", 5)
        out[name] = (int(n), float(ms), float(lo), float(med), float(hi)) if spread else (int(n), float(ms))
    return out


def copy_probe(nbytes=1 << 30, reps=5):
    """Measured HBM rates of this device (upsp_copy_probe: streaming float4 copy and fill of `nbytes`), GB/s."""
    import torch
    a = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
    b = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
    a.zero_()
    ms = C.c_float()
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    check(lib().upsp_copy_probe(C.c_void_p(a.data_ptr()), C.c_void_p(b.data_ptr()), nbytes, reps, C.byref(ms), st))
    copy = 2.0 * nbytes / (ms.value * 1e-3) / 1e9
    check(lib().upsp_copy_probe(None, C.c_void_p(b.data_ptr()), nbytes, reps, C.byref(ms), st))
    fill = nbytes / (ms.value * 1e-3) / 1e9
    return {"copy_GBps": copy, "fill_GBps": fill, "bytes": int(nbytes), "reps": int(reps),
            "kernel": "upsp::copy_probe_kernel / fill_probe_kernel (float4 per lane; fastest of non-temporal / plain at 8, 16, 32 workgroups per CU)"}


def bandwidth_probe(nbytes=1 << 30, reps=5):
    """Measured read-only and write-only HBM rates in the shapes of the frame loop's two passes (upsp_bandwidth_probe), GB/s."""
    import torch
    a = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
    a.zero_()
    ms = C.c_float()
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    out = {"bytes": int(nbytes), "reps": int(reps)}
    for kind, name in ((0, "read_GBps"), (1, "write_GBps")):
        check(lib().upsp_bandwidth_probe(kind, C.c_void_p(a.data_ptr()), nbytes, reps, C.byref(ms), st))
        out[name] = nbytes / (ms.value * 1e-3) / 1e9
    return out


def comm_library():
    """Path of the RCCL build the library bound (upsp_comm_library)."""
    buf = C.create_string_buffer(4096)
    check(lib().upsp_comm_library(buf, len(buf)))
    return buf.value.decode()
