"""ctypes binding of the CPU ORACLE (test infrastructure, NOT product code).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module.  See oracle/upsp_oracle.h for the reference file:line each function
restates and for the parity status (pinned / unpinned) of each group.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libupsp_oracle.so")


def build(force=False):
    """Compile oracle/*.c with gcc (recipe: oracle/Makefile)."""
    srcs = [f for f in os.listdir(_HERE) if f.endswith("_oracle.c") or f.endswith(".h")]
    stale = (not os.path.exists(_LIB_PATH)) or any(
        os.path.getmtime(os.path.join(_HERE, s)) > os.path.getmtime(_LIB_PATH) for s in srcs
    )
    if force or stale:
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _LIB_PATH


class Ray(C.Structure):
    _fields_ = [("o", C.c_float * 3), ("d", C.c_float * 3), ("inv", C.c_float * 3),
                ("kx", C.c_int), ("ky", C.c_int), ("kz", C.c_int),
                ("Sx", C.c_float), ("Sy", C.c_float), ("Sz", C.c_float)]


class Hit(C.Structure):
    _fields_ = [("pos", C.c_float * 3), ("nrm", C.c_float * 3),
                ("t", C.c_float), ("u", C.c_float), ("v", C.c_float), ("w", C.c_float),
                ("geomID", C.c_int32), ("primID", C.c_int32)]


class Node(C.Structure):
    _fields_ = [("bmin", C.c_float * 3), ("bmax", C.c_float * 3), ("offset", C.c_int32),
                ("nprims", C.c_uint16), ("axis", C.c_uint8), ("pad", C.c_uint8)]


class BvhStruct(C.Structure):
    _fields_ = [("ntris", C.c_size_t), ("verts", C.POINTER(C.c_float)),
                ("prim_ids", C.POINTER(C.c_int32)), ("nodes", C.POINTER(Node)),
                ("nnodes", C.c_int32), ("depth", C.c_int32)]


class Camera(C.Structure):
    _fields_ = [("K", C.c_double * 9), ("dist", C.c_double * 5), ("R", C.c_double * 9),
                ("t", C.c_double * 3), ("width", C.c_int32), ("height", C.c_int32)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB_PATH)
        L.orc_bvh_create.restype = C.POINTER(BvhStruct)
        L.orc_bvh_create.argtypes = [C.c_void_p, C.c_size_t]
        L.orc_bvh_destroy.argtypes = [C.POINTER(BvhStruct)]
        L.orc_bvh_intersect.restype = C.c_int
        L.orc_bvh_intersect.argtypes = [C.POINTER(BvhStruct), C.POINTER(Ray), C.POINTER(Hit),
                                        C.c_void_p, C.c_void_p]
        L.orc_bvh_intersect_batch.argtypes = [C.POINTER(BvhStruct), C.c_void_p, C.c_void_p,
                                              C.c_size_t, C.c_int] + [C.c_void_p] * 6 + \
                                             [C.c_int, C.c_void_p, C.c_void_p]
        L.orc_ray_init.argtypes = [C.POINTER(Ray), C.c_void_p, C.c_void_p]
        L.orc_hit_init.argtypes = [C.POINTER(Hit)]
        L.orc_project_point.argtypes = [C.POINTER(Camera), C.c_void_p, C.c_void_p]
        L.orc_cam_center.argtypes = [C.POINTER(Camera), C.c_void_p]
        L.orc_create_projection.restype = C.c_int64
        L.orc_create_projection.argtypes = [C.POINTER(BvhStruct), C.POINTER(Camera)] + \
            [C.c_void_p] * 4 + [C.c_size_t, C.c_float] + [C.c_void_p] * 4 + [C.c_int]
        L.orc_adjust_weights.argtypes = [C.c_int, C.c_size_t] + [C.c_void_p] * 5 + [C.c_int]
        L.orc_skipped_nodes.restype = C.c_size_t
        L.orc_skipped_nodes.argtypes = [C.c_int, C.c_size_t, C.c_void_p, C.c_void_p]
        L.orc_fix_hot_pixels.restype = C.c_int
        L.orc_fix_hot_pixels.argtypes = [C.c_void_p] + [C.c_int] * 5
        L.orc_project_frame_f32.argtypes = [C.c_void_p] * 3 + [C.c_size_t, C.c_void_p]
        L.orc_project_frame_u16.argtypes = [C.c_void_p] * 3 + [C.c_size_t, C.c_void_p]
        L.orc_accumulate.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]
        L.orc_frame_loop_u16.argtypes = ([C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                          C.c_size_t, C.c_size_t, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                          C.c_void_p, C.c_int, C.c_void_p])
        L.orc_frame_loop_u16.restype = None
        L.orc_finals.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_uint64,
                                 C.c_void_p, C.c_void_p]
        L.orc_apportion.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        L.orc_transpose.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
        for name, args, res in _OPTIONAL:
            if hasattr(L, name):
                getattr(L, name).argtypes = args
                getattr(L, name).restype = res
        _lib = L
    return _lib


_OPTIONAL = [
    ("orc_unpack_12bit", [C.c_void_p, C.c_size_t, C.c_void_p], None),
    ("orc_unpack_10bit", [C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p], None),
    ("orc_gaussian_kernel", [C.c_int, C.c_void_p], C.c_int),
    ("orc_blur_f32", [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int], None),
    ("orc_warp_affine_u16", [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int], None),
    ("orc_warp_affine_f32", [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int], None),
    ("orc_find_transform_ecc", [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int,
                                C.c_double, C.c_void_p], C.c_int),
    ("orc_register_pixel_u16", [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int,
                                C.c_double, C.c_int, C.c_void_p], C.c_int),
    ("orc_polyfit2d", [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p], C.c_int),
    ("orc_polyval2d", [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p], None),
    ("orc_patch_clusters", [C.c_void_p, C.c_int, C.c_int] + [C.c_void_p] * 6, None),
    ("orc_interpolate_idw", [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_int, C.c_float,
                             C.c_void_p, C.c_void_p], None),
    ("orc_kd_build", [C.c_void_p, C.c_size_t], C.c_void_p),
    ("orc_kd_free", [C.c_void_p], None),
    ("orc_kd_nearest_batch", [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p], None),
    ("orc_get_targets", [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_float,
                         C.c_void_p], None),
    ("orc_target_diameters", [C.c_void_p] * 6 + [C.c_size_t, C.c_void_p], None),
    ("orc_cluster_points", [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p], C.c_int),
    ("orc_intensity_histc", [C.c_void_p, C.c_size_t, C.c_uint, C.c_int, C.c_void_p, C.c_void_p], None),
    ("orc_find_peaks", [C.c_void_p, C.c_int, C.c_uint, C.c_void_p], C.c_int),
    ("orc_first_min_threshold", [C.c_void_p, C.c_int, C.c_uint], C.c_uint),
    ("orc_patch_tables", [C.c_void_p] * 4 + [C.c_int, C.c_int, C.c_int, C.c_uint, C.c_uint, C.c_void_p,
                          C.c_uint, C.c_uint] + [C.c_void_p] * 6, None),
    ("orc_free", [C.c_void_p], None),
    ("orc_transpoly_design", [C.c_int, C.c_int, C.c_void_p], None),
    ("orc_transpoly_fit", [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p], C.c_int),
    ("orc_paint_gain", [C.c_void_p, C.c_float, C.c_float], C.c_float),
    ("orc_phase2", [C.c_void_p, C.c_size_t, C.c_int] + [C.c_void_p] * 5 + [C.c_float, C.c_float, C.c_int]
     + [C.c_void_p] * 4 + [C.c_int], None),
]


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def make_camera(K, dist, R, t, width, height):
    cam = Camera()
    cam.K[:] = np.asarray(K, dtype=np.float64).ravel().tolist()
    d = np.zeros(5)
    dd = np.asarray(dist, dtype=np.float64).ravel()
    d[: min(5, dd.size)] = dd[:5]
    cam.dist[:] = d.tolist()
    cam.R[:] = np.asarray(R, dtype=np.float64).ravel().tolist()
    cam.t[:] = np.asarray(t, dtype=np.float64).ravel().tolist()
    cam.width, cam.height = int(width), int(height)
    return cam


class OracleBVH:
    """rt::BVH restated on the CPU (reference: cpp/raycast/pspRT.cpp)."""

    def __init__(self, tris9):
        tris9 = _f32(tris9).reshape(-1)
        assert tris9.size % 9 == 0
        self.ntris = tris9.size // 9
        self._h = lib().orc_bvh_create(_p(tris9), self.ntris)
        self._destroy = lib().orc_bvh_destroy
        if not self._h:
            raise ValueError("BVH::BVH() : no primitives!")

    def __del__(self):
        if getattr(self, "_h", None):
            self._destroy(self._h)
            self._h = None

    @property
    def nnodes(self):
        return self._h.contents.nnodes

    @property
    def depth(self):
        return self._h.contents.depth

    def nodes(self):
        n = self.nnodes
        buf = C.string_at(self._h.contents.nodes, n * C.sizeof(Node))
        dt = np.dtype([("bmin", "<f4", 3), ("bmax", "<f4", 3), ("offset", "<i4"),
                       ("nprims", "<u2"), ("axis", "u1"), ("pad", "u1")])
        return np.frombuffer(buf, dtype=dt).copy()

    def prim_ids(self):
        return np.ctypeslib.as_array(self._h.contents.prim_ids, shape=(self.ntris,)).copy()

    def intersect_one(self, o, d):
        r, h = Ray(), Hit()
        lib().orc_ray_init(C.byref(r), _p(_f32(o)), _p(_f32(d)))
        lib().orc_hit_init(C.byref(h))
        any_hit = lib().orc_bvh_intersect(self._h, C.byref(r), C.byref(h), None, None)
        return bool(any_hit), h

    def intersect(self, org, dirs, threads=0, stats=False):
        """Batch closest hit.  org: (3,) shared or (N,3); dirs: (N,3)."""
        dirs = _f32(dirs).reshape(-1, 3)
        org = _f32(org)
        n = dirs.shape[0]
        stride = 0 if org.size == 3 else 3
        if stride:
            assert org.reshape(-1, 3).shape[0] == n
        hit = np.zeros(n, np.uint8)
        t = np.zeros(n, np.float32)
        prim = np.zeros(n, np.int32)
        uvw = np.zeros((n, 3), np.float32)
        pos = np.zeros((n, 3), np.float32)
        nrm = np.zeros((n, 3), np.float32)
        nv, nt = C.c_uint64(0), C.c_uint64(0)
        lib().orc_bvh_intersect_batch(self._h, _p(org), _p(dirs), n, stride, _p(hit), _p(t),
                                      _p(prim), _p(uvw), _p(pos), _p(nrm), int(threads),
                                      C.addressof(nv), C.addressof(nt))
        out = dict(hit=hit.astype(bool), t=t, prim=prim, uvw=uvw, pos=pos, nrm=nrm)
        if stats:
            out["nodes_visited"] = nv.value
            out["tris_tested"] = nt.value
        return out


def project_points(cam, xyz):
    xyz = _f32(xyz).reshape(-1, 3)
    out = np.zeros((xyz.shape[0], 2), np.float32)
    for i in range(xyz.shape[0]):
        lib().orc_project_point(C.byref(cam), _p(xyz[i]), _p(out[i]))
    return out


def cam_center(cam):
    c = np.zeros(3, np.float64)
    lib().orc_cam_center(C.byref(cam), _p(c))
    return c


def create_projection(bvh, cam, nodes, normals, tri_nodes, oblique_thresh, datanode=None,
                      threads=0):
    nodes = _f32(nodes).reshape(-1, 3)
    normals = _f32(normals).reshape(-1, 3)
    tri_nodes = np.ascontiguousarray(tri_nodes, dtype=np.int32).reshape(-1)
    n = nodes.shape[0]
    pix = np.zeros(n, np.int32)
    uv = np.zeros(2 * n, np.float32)
    cnt = np.zeros(cam.width * cam.height, np.uint8)
    nrays = C.c_uint64(0)
    dn = None if datanode is None else np.ascontiguousarray(datanode, dtype=np.uint8)
    acc = lib().orc_create_projection(bvh._h, C.byref(cam), _p(nodes), _p(normals), _p(dn),
                                      _p(tri_nodes), n, np.float32(oblique_thresh), _p(pix),
                                      _p(uv), _p(cnt), C.addressof(nrays), int(threads))
    return dict(pix=pix, uv=uv, nodecount=cnt.reshape(cam.height, cam.width),
                accepted=int(acc), nrays=int(nrays.value))


def oblique_ambiguous(cam, nodes, normals, oblique_thresh, datanode=None):
    """In-frame nodes whose oblique verdict depends on the last bit of libm's acosf (orc_oblique_ambiguous)."""
    nodes = _f32(nodes).reshape(-1, 3)
    normals = _f32(normals).reshape(-1, 3)
    dn = None if datanode is None else np.ascontiguousarray(datanode, dtype=np.uint8)
    L = lib()
    L.orc_oblique_ambiguous.restype = C.c_int64
    L.orc_oblique_ambiguous.argtypes = [C.POINTER(Camera), C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_float]
    return int(L.orc_oblique_ambiguous(C.byref(cam), _p(nodes), _p(normals), _p(dn), nodes.shape[0],
                                       np.float32(oblique_thresh)))


def adjust_weights(pix, weight, nodes, normals, centers, mode):
    pix = np.ascontiguousarray(pix, dtype=np.int32)
    ncams, n = pix.shape
    weight = np.ascontiguousarray(weight, dtype=np.float32).copy()
    centers = np.ascontiguousarray(centers, dtype=np.float64)
    lib().orc_adjust_weights(ncams, n, _p(pix), _p(weight), _p(_f32(nodes)), _p(_f32(normals)),
                             _p(centers), int(mode))
    return weight


def skipped_nodes(pix):
    pix = np.ascontiguousarray(pix, dtype=np.int32)
    if pix.ndim == 1:
        pix = pix[None]
    ncams, n = pix.shape
    sk = np.zeros(n, np.uint8)
    lib().orc_skipped_nodes(ncams, n, _p(pix), _p(sk))
    return sk.astype(bool)


def fix_hot_pixels(img, thresh=4064, min_change=512, max_hot=5):
    """Returns (fixed copy, status) ; status -1 = too many hot pixels (frame untouched)."""
    img = np.ascontiguousarray(img, dtype=np.uint16).copy()
    st = lib().orc_fix_hot_pixels(_p(img), img.shape[0], img.shape[1], thresh, min_change, max_hot)
    return img, st


def project_frame(img, pix, weight=None):
    pix = np.ascontiguousarray(pix, dtype=np.int32)
    out = np.zeros(pix.size, np.float32)
    w = None if weight is None else _f32(weight)
    img = np.ascontiguousarray(img)
    if img.dtype == np.uint16:
        lib().orc_project_frame_u16(_p(img), _p(pix), _p(w), pix.size, _p(out))
    else:
        img = _f32(img)
        lib().orc_project_frame_f32(_p(img), _p(pix), _p(w), pix.size, _p(out))
    return out


def accumulate(sol, s, ss):
    lib().orc_accumulate(_p(_f32(sol)), sol.size, _p(s), _p(ss))


def frame_loop(frames, pix, weight=None, skipped=None, thresh=4064, min_change=512, max_hot=5,
               want_rows=True, threads=1, timing=None):
    """The plain frame loop of psp_process.cpp:1742-1851 (OpenMP over frames, thread-private double
    accumulators).  frames: u16 [F,H,W], repaired IN PLACE.  Returns (rows [F,N] or None, sum, sumsq).
    timing: dict that receives the seconds spent in set-up / frame loop / merge."""
    assert frames.dtype == np.uint16 and frames.flags.c_contiguous and frames.ndim == 3
    pix = np.ascontiguousarray(pix, dtype=np.int32)
    w = None if weight is None else _f32(weight)
    sk = np.nonzero(skipped_nodes(pix) if skipped is None else np.asarray(skipped))[0].astype(np.int32)
    F, H, W = frames.shape
    rows = np.empty((F, pix.size), np.float32) if want_rows else None
    s, ss = np.zeros(pix.size), np.zeros(pix.size)
    sec = None if timing is None else np.zeros(3)
    lib().orc_frame_loop_u16(_p(frames), F, H, W, _p(pix), _p(w), _p(sk), sk.size, pix.size, thresh, min_change,
                             max_hot, _p(rows), _p(s), _p(ss), int(threads), _p(sec))
    if timing is not None:
        timing.update(setup=float(sec[0]), loop=float(sec[1]), merge=float(sec[2]))
    return rows, s, ss


def finals(s, ss, nframes):
    avg = np.zeros(s.size, np.float32)
    rms = np.zeros(s.size, np.float32)
    lib().orc_finals(_p(s), _p(ss), s.size, int(nframes), _p(avg), _p(rms))
    return avg, rms


def apportion(value, nbins):
    st = np.zeros(nbins, np.int32)
    ex = np.zeros(nbins, np.int32)
    lib().orc_apportion(int(value), int(nbins), _p(st), _p(ex))
    return st, ex


def transpose(src):
    src = _f32(src)
    y, x = src.shape
    dst = np.zeros((x, y), np.float32)
    lib().orc_transpose(_p(src), x, y, _p(dst))
    return dst


# ---- image operators (PARITY UNPINNED: OpenCV / Eigen restatements) --------------

def blur(img, k, box=False):
    img = _f32(img)
    out = np.zeros_like(img)
    lib().orc_blur_f32(_p(img), _p(out), img.shape[0], img.shape[1], int(k), int(bool(box)))
    return out


def warp_affine(img, M, interp=1):
    M = _f32(M).reshape(6)
    img = np.ascontiguousarray(img)
    out = np.zeros_like(img)
    if img.dtype == np.uint16:
        lib().orc_warp_affine_u16(_p(img), _p(out), img.shape[0], img.shape[1], _p(M), int(interp))
    else:
        img = _f32(img)
        out = np.zeros_like(img)
        lib().orc_warp_affine_f32(_p(img), _p(out), img.shape[0], img.shape[1], _p(M), int(interp))
    return out


def find_transform_ecc(ref, inp, M=None, max_iters=50, eps=1e-3):
    ref, inp = _f32(ref), _f32(inp)
    M = np.array([1, 0, 0, 0, 1, 0], np.float32) if M is None else _f32(M).reshape(6).copy()
    rho = C.c_double(0)
    it = lib().orc_find_transform_ecc(_p(ref), _p(inp), ref.shape[0], ref.shape[1], _p(M),
                                      int(max_iters), float(eps), C.addressof(rho))
    return M.reshape(2, 3), it, rho.value


def register_pixel(ref, frame_u16, max_iters=50, eps=1e-3, interp=1):
    ref = _f32(ref)
    fr = np.ascontiguousarray(frame_u16, dtype=np.uint16)
    M = np.zeros(6, np.float32)
    out = np.zeros_like(fr)
    it = lib().orc_register_pixel_u16(_p(ref), _p(fr), fr.shape[0], fr.shape[1], _p(M),
                                      int(max_iters), float(eps), int(interp), _p(out))
    return out, M.reshape(2, 3), it


def polyfit2d(x, y, z):
    x = np.ascontiguousarray(x, dtype=np.int32)
    y = np.ascontiguousarray(y, dtype=np.int32)
    z = _f32(z)
    poly = np.zeros(10, np.float32)
    rank = lib().orc_polyfit2d(_p(x), _p(y), _p(z), x.size, _p(poly))
    return poly, rank


def patch_clusters(img, clusters):
    img = _f32(img).copy()
    b_off, i_off = [0], [0]
    bx, by, ix, iy = [], [], [], []
    for cl in clusters:
        bx += list(cl["bx"]); by += list(cl["by"]); ix += list(cl["ix"]); iy += list(cl["iy"])
        b_off.append(len(bx)); i_off.append(len(ix))
    arr = [np.asarray(a, dtype=np.int32) for a in (b_off, bx, by, i_off, ix, iy)]
    lib().orc_patch_clusters(_p(img), img.shape[1], len(clusters), *[_p(a) for a in arr])
    return img


def unpack_12bit(packed):
    packed = np.ascontiguousarray(packed, dtype=np.uint8).reshape(-1)
    out = np.zeros(packed.size * 2 // 3, np.uint16)
    lib().orc_unpack_12bit(_p(packed), packed.size, _p(out))
    return out


def unpack_10bit(packed, lut=None):
    packed = np.ascontiguousarray(packed, dtype=np.uint8).reshape(-1)
    out = np.zeros(packed.size * 4 // 5, np.uint16)
    lut = None if lut is None else np.ascontiguousarray(lut, dtype=np.uint16)
    lib().orc_unpack_10bit(_p(packed), packed.size, _p(lut), _p(out))
    return out


def transpoly_design(nframes, degree):
    """TransPolyFitter design matrix, returned [nframes, degree+1] (A_(f,c))."""
    A = np.zeros((degree + 1, nframes), np.float32)
    lib().orc_transpoly_design(nframes, degree, _p(A))
    return A.T.copy()


def transpoly_fit(y, degree, nframes=None):
    """TransPolyFitter::eval_fit for one point: returns (poly[degree+1], fit[nframes])."""
    y = _f32(y).reshape(-1)
    n = y.size if nframes is None else nframes
    A = np.zeros((degree + 1, n), np.float32)
    lib().orc_transpoly_design(n, degree, _p(A))
    poly = np.zeros(degree + 1, np.float32)
    fit = np.zeros(n, np.float32)
    lib().orc_transpoly_fit(_p(A), n, degree + 1, _p(y), _p(poly), _p(fit))
    return poly, fit


def paint_gain(cal, T, Pss):
    cal = _f32(cal).reshape(6)
    return float(lib().orc_paint_gain(_p(cal), float(T), float(Pss)))


def phase2(intensity_t, iref, coverage, steady, model_temp, cal, qbar, ps, degree=6, threads=0):
    """psp_process phase-2 node loop.  Returns dict(pressure_t, sum, sumsq, gain)."""
    I = _f32(intensity_t)
    n, F = I.shape
    out = np.full((n, F), np.nan, np.float32)
    s = np.zeros(n, np.float64)
    ss = np.zeros(n, np.float64)
    g = np.zeros(n, np.float64)
    cal = _f32(cal).reshape(6)
    iref, coverage, steady, model_temp = (_f32(a).reshape(n) for a in (iref, coverage, steady, model_temp))
    lib().orc_phase2(_p(I), n, F, _p(iref), _p(coverage), _p(steady), _p(model_temp), _p(cal),
                     float(qbar), float(ps), int(degree), _p(out), _p(s), _p(ss), _p(g), int(threads))
    return dict(pressure_t=out, sum=s, sumsq=ss, gain=g)


class KdTree:
    """kd-tree over the model nodes, restated (oracle/kd_oracle.c)."""

    def __init__(self, nodes):
        nodes = _f32(nodes).reshape(-1, 3)
        self._free = lib().orc_kd_free
        self._h = lib().orc_kd_build(_p(nodes), nodes.shape[0])

    def nearest(self, queries):
        q = np.ascontiguousarray(queries, dtype=np.float64).reshape(-1, 3)
        idx = np.zeros(q.shape[0], np.int32)
        d2 = np.zeros(q.shape[0], np.float64)
        lib().orc_kd_nearest_batch(self._h, _p(q), q.shape[0], _p(idx), _p(d2))
        return idx, d2

    def __del__(self):
        if getattr(self, "_h", None):
            self._free(self._h)
            self._h = None


_REF_KD = os.path.join(_HERE, "_ref", "libpspkdtree.so")


def build_ref():
    """Compile the buildable part of the reference (the vendored kd-tree) into oracle/_ref/
    when /root/reference is present; returns True if the library exists afterwards."""
    if os.path.isdir("/root/reference/cpp/raycast"):
        subprocess.check_call(["make", "-C", _HERE, "-s", "ref"])
    return os.path.exists(_REF_KD)


class RefKdTree:
    """The reference's own kd-tree (cpp/raycast/pspKdtree.c compiled into oracle/_ref/),
    driven like TriModel_::generate_kd_tree (cpp/lib/TriModel.ipp:915-937) and getTargets
    (cpp/exec/psp_process.cpp:95-100)."""

    def __init__(self, nodes):
        if not os.path.exists(_REF_KD):
            raise FileNotFoundError(_REF_KD)
        L = C.CDLL(_REF_KD)
        L.kd_create.restype = C.c_void_p
        L.kd_create.argtypes = [C.c_int]
        L.kd_insert3.argtypes = [C.c_void_p, C.c_double, C.c_double, C.c_double, C.c_void_p]
        L.kd_nearest.restype = C.c_void_p
        L.kd_nearest.argtypes = [C.c_void_p, C.c_void_p]
        L.kd_res_item_data.restype = C.c_void_p
        L.kd_res_item_data.argtypes = [C.c_void_p]
        L.kd_res_free.argtypes = [C.c_void_p]
        L.kd_free.argtypes = [C.c_void_p]
        self.L = L
        self.t = L.kd_create(3)
        nodes = _f32(nodes).reshape(-1, 3)
        for i in range(nodes.shape[0]):
            # user data = node index + 1 (a NULL pointer cannot be told from index 0 via ctypes)
            L.kd_insert3(self.t, float(nodes[i, 0]), float(nodes[i, 1]), float(nodes[i, 2]), C.c_void_p(i + 1))

    def nearest(self, queries):
        q = np.ascontiguousarray(queries, dtype=np.float64).reshape(-1, 3)
        out = np.zeros(q.shape[0], np.int32)
        for k in range(q.shape[0]):
            res = self.L.kd_nearest(self.t, q[k].ctypes.data_as(C.c_void_p))
            out[k] = int(self.L.kd_res_item_data(res)) - 1
            self.L.kd_res_free(res)
        return out

    def __del__(self):
        if getattr(self, "t", None):
            self.L.kd_free(self.t)
            self.t = None


# ------------------------------------------------------------- phase-0 patch set-up --
def get_targets(bvh, kd, cam, normals, xyz, oblique_thresh):
    xyz = _f32(xyz).reshape(-1, 3)
    normals = _f32(normals).reshape(-1, 3)
    keep = np.zeros(xyz.shape[0], np.uint8)
    lib().orc_get_targets(bvh._h, kd._h, C.byref(cam), _p(normals), _p(xyz), xyz.shape[0],
                          float(oblique_thresh), _p(keep))
    return keep.astype(bool)


def target_diameters(kd, cam, normals, xyz, uv, diam):
    xyz, uv, diam = _f32(xyz).reshape(-1, 3), _f32(uv).reshape(-1, 2), _f32(diam).reshape(-1)
    normals = _f32(normals).reshape(-1, 3)
    out = np.zeros(diam.size, np.float32)
    lib().orc_target_diameters(kd._h, C.byref(cam), _p(normals), _p(xyz), _p(uv), _p(diam), diam.size, _p(out))
    return out


def cluster_points(uv, diam, bound_pts):
    uv, diam = _f32(uv).reshape(-1, 2), _f32(diam).reshape(-1)
    n = diam.size
    order = np.zeros(max(n, 1), np.int32)
    off = np.zeros(n + 1, np.int32)
    ncl = lib().orc_cluster_points(_p(uv), _p(diam), n, int(bound_pts), _p(order), _p(off))
    return order[:n], off[:ncl + 1]


def intensity_histc(img, depth=12, bins=-1):
    img = np.ascontiguousarray(img, dtype=np.uint16)
    nb = (1 << min(depth, 16)) if bins == -1 else bins
    edges = np.zeros(nb + 1, np.int32)
    counts = np.zeros(nb, np.int32)
    lib().orc_intensity_histc(_p(img), img.size, depth, bins, _p(edges), _p(counts))
    return edges, counts


def find_peaks(data, separation=0):
    d = np.ascontiguousarray(data, dtype=np.float64)
    peaks = np.zeros(max(d.size, 1), np.uint32)
    n = lib().orc_find_peaks(_p(d), d.size, separation, _p(peaks))
    return peaks[:n].tolist()


def first_min_threshold(counts, separation=1):
    c = np.ascontiguousarray(counts, dtype=np.int32)
    return int(lib().orc_first_min_threshold(_p(c), c.size, separation))


def patch_tables(uv, diam, order, cl_off, size, bound_pts=2, buffer=1, ref=None, thresh=0, offset=2):
    """PatchClusters ctor (+ threshold_bounds when ref is given) -> list of dict(bx,by,ix,iy)."""
    uv, diam = _f32(uv).reshape(-1, 2), _f32(diam).reshape(-1)
    order = np.ascontiguousarray(order, dtype=np.int32)
    cl_off = np.ascontiguousarray(cl_off, dtype=np.int32)
    ncl = cl_off.size - 1
    refa = None if ref is None else np.ascontiguousarray(ref, dtype=np.uint16)
    ptrs = [C.c_void_p() for _ in range(6)]
    lib().orc_patch_tables(_p(uv), _p(diam), _p(order), _p(cl_off), ncl, size[0], size[1], bound_pts,
                           buffer, _p(refa), int(thresh), int(offset), *[C.byref(p) for p in ptrs])

    def take(ptr, n):
        if n == 0 or not ptr.value:
            return np.zeros(0, np.int32)
        return np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_int32)), shape=(n,)).copy()
    b_off = take(ptrs[0], ncl + 1)
    i_off = take(ptrs[3], ncl + 1)
    bx, by = take(ptrs[1], int(b_off[-1])), take(ptrs[2], int(b_off[-1]))
    ix, iy = take(ptrs[4], int(i_off[-1])), take(ptrs[5], int(i_off[-1]))
    for p in ptrs:
        if p.value:
            lib().orc_free(p)
    return [dict(bx=bx[b_off[c]:b_off[c + 1]], by=by[b_off[c]:b_off[c + 1]],
                 ix=ix[i_off[c]:i_off[c + 1]], iy=iy[i_off[c]:i_off[c + 1]]) for c in range(ncl)]


def interpolate_idw(src_nodes, src_data, query_nodes, k=10, p=2.0):
    """upsp::interpolate restated exhaustively: returns (values f32 [Q], neighbours int32 [Q,k])."""
    src = _f32(src_nodes).reshape(-1, 3)
    data = _f32(src_data).reshape(-1)
    q = _f32(query_nodes).reshape(-1, 3)
    out = np.zeros(q.shape[0], np.float32)
    nbr = np.zeros((q.shape[0], k), np.int32)
    lib().orc_interpolate_idw(_p(src), _p(data), src.shape[0], _p(q), q.shape[0], int(k), float(p), _p(out), _p(nbr))
    return out, nbr
