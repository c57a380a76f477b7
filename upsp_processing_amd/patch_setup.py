"""Phase-0 patch set-up of psp_process: from a target file to the per-camera patch tables
the frame pipeline consumes (InitializeImagePatches, cpp/exec/psp_process.cpp:2088-2182).

    read_psp_target_file -> getTargets (visibility: GPU ray cast + nearest node + oblique test)
    -> map to image -> get_target_diameters -> cluster_points -> PatchClusters (boundary /
    interior pixel lists) -> histogram threshold of the first frame -> threshold_bounds

The geometry queries run on the GPU engine (BVH closest hit, nearest node, projectPoints in
the library); everything else is small integer / pixel-list logic on a few dozen targets and
stays on the host, written to reproduce the reference's arithmetic (float / double mix,
rounding at floor / ceil, iteration order, and the early `break` of find_peaks).
"""
import math

import numpy as np

from . import engine

f32 = np.float32
PI = 3.141592653589793          # cpp/include/utils/general_utils.h:17-18


class Target:
    """upsp::Target_<float> (cpp/include/data_structs.h:50-60)."""
    __slots__ = ("xyz", "uv", "diameter", "num")

    def __init__(self, xyz=(0, 0, 0), uv=(0, 0), diameter=0.0, num=0):
        self.xyz = np.array(xyz, dtype=np.float32)
        self.uv = np.array(uv, dtype=np.float32)
        self.diameter = f32(diameter)
        self.num = int(num)

    def copy(self):
        return Target(self.xyz, self.uv, self.diameter, self.num)


def read_psp_target_file(path, label="*Targets", planar=False):
    """upsp_files::read_psp_target_file (cpp/utils/file_readers.ipp:207-254): the block that
    follows the first line containing `label`, up to the next line starting with '*'.
    Columns: id x y z nx ny nz diameter ...  (stream extraction: a short line leaves zeros)."""
    targs = []
    with open(path) as f:
        lines = f.read().split("\n")
    if lines and lines[-1] == "":
        lines.pop()
    # x, y, z, diam live outside the line loop in the reference: after the first field that
    # fails to parse (which is zeroed, C++11 extraction) the rest keep the previous line's values
    x = y = z = diam = 0.0
    i = 0
    while i < len(lines):
        if label in lines[i]:
            i += 1
            while i < len(lines):
                line = lines[i]
                if line[:1] == "*":
                    break
                toks = line.split()
                vals = {"id": 0, 1: x, 2: y, 3: z, 7: diam}
                ok = True
                for k in range(8):
                    if not ok:
                        break
                    try:
                        v = (int(toks[k]) if k == 0 else float(toks[k]))
                    except (IndexError, ValueError):
                        ok = False
                        v = 0
                    if k == 0:
                        vals["id"] = v
                    elif k in (1, 2, 3, 7):
                        vals[k] = float(v)
                x, y, z, diam = vals[1], vals[2], vals[3], vals[7]
                if planar:
                    z = 0.0
                targs.append(Target((x, y, z), (0, 0), diam, vals["id"]))
                i += 1
            break
        i += 1
    return targs


def cv_round(v):
    """cvRound: round half to even; NaN and values beyond the int range give INT_MIN like cvtss2si."""
    v = float(v)
    if not (-2147483648.0 <= v < 2147483648.0):
        return -2147483648
    return int(np.rint(v))


def contains(size, uv):
    """upsp::contains(cv::Size, cv::Point2i) on a float point (cpp/lib/projection.cpp:10-13;
    Point2f -> Point2i rounds)."""
    x, y = cv_round(uv[0]), cv_round(uv[1])
    return 0 <= x < size[0] and 0 <= y < size[1]


def target_oblique_threshold(oblique_angle):
    """deg2_rad(180. - min(oblique_angle + 5.0, 90.0)) narrowed to float (psp_process.cpp:2108)."""
    return float(f32((180.0 - min(float(f32(oblique_angle)) + 5.0, 90.0)) * PI / 180.0))


def _v3_length(v):
    # Imath::Vec3<float>::length(): sqrt(x*x + y*y + z*z) in float
    return f32(np.sqrt(f32(f32(f32(v[0] * v[0]) + f32(v[1] * v[1])) + f32(v[2] * v[2]))))


def get_targets(bvh, cam, size, targets, d_nodes, normals, oblique_thresh):
    """getTargets (psp_process.cpp:55-112): keeps the targets that project into the frame,
    are not occluded (closest hit no nearer than dist - 1e-3) and face the camera
    (acos(n . dir) > thresh with n = normal of the model node nearest to the hit)."""
    if not targets:
        return []
    center = engine.camera_center(cam)
    orig = np.array([f32(center[0]), f32(center[1]), f32(center[2])], np.float32)
    xyz = np.stack([t.xyz for t in targets]).astype(np.float32)
    uv = engine.project_points(cam, xyz)
    cand, dirs, dist = [], [], []
    for i, t in enumerate(targets):
        u, v = uv[i]
        if u < 0 or v < 0 or u >= size[0] or v >= size[1]:          # :74-77 (float compare)
            continue
        d = (t.xyz - orig).astype(np.float32)
        l = _v3_length(d)
        if l != 0:
            d = (d / l).astype(np.float32)                            # V3f::normalize
        cand.append(i)
        dirs.append(d)
        dist.append(l)
    if not cand:
        return []
    dirs = np.stack(dirs)
    h = bvh.intersect(orig, dirs, want=("hit", "t", "pos"))
    hit = h["hit"].cpu().numpy()
    tt = h["t"].cpu().numpy()
    pos = h["pos"].cpu().numpy()
    keep = [k for k in range(len(cand)) if hit[k] and not (float(tt[k]) < float(dist[k]) - 1e-3)]   # :90
    if not keep:
        return []
    near = engine.nearest_nodes(d_nodes, pos[keep].astype(np.float64)).cpu().numpy()
    normals = np.asarray(normals, dtype=np.float32).reshape(-1, 3)
    out = []
    for j, k in enumerate(keep):
        n, d = normals[near[j]], dirs[k]
        cos_t = f32(f32(f32(n[0] * d[0]) + f32(n[1] * d[1])) + f32(n[2] * d[2]))
        c = float(cos_t)
        ang = f32(math.acos(c)) if -1.0 <= c <= 1.0 else f32("nan")     # acos in double, narrowed
        if ang > f32(oblique_thresh):                                 # :102-108
            out.append(targets[cand[k]].copy())
    return out


def map_points_to_image(cam, targets):
    """CameraCal::map_points_to_image (cpp/lib/CameraCal.ipp:204-224)."""
    if targets:
        uv = engine.project_points(cam, np.stack([t.xyz for t in targets]))
        for t, p in zip(targets, uv):
            t.uv = p.astype(np.float32)


def get_perpendicular(vec):
    """upsp::get_perpendicular (cpp/utils/cv_extras.ipp:29-66), float."""
    v = np.asarray(vec, dtype=np.float32)
    norm = f32(np.sqrt(float(v[0]) * float(v[0]) + float(v[1]) * float(v[1]) + float(v[2]) * float(v[2])))
    out = np.zeros(3, np.float32)
    if norm == 0:
        return out
    v = (v / norm).astype(np.float32)
    a = np.abs(v)
    m = (0 if a[0] > a[2] else 2) if a[0] > a[1] else (1 if a[1] > a[2] else 2)   # max_ind
    if m == 0:
        out[1] = 1.0
        out[0] = -f32(f32(out[1] * v[1]) + f32(out[2] * v[2])) / v[0]
    elif m == 1:
        out[0] = 1.0
        out[1] = -f32(f32(out[0] * v[0]) + f32(out[2] * v[2])) / v[1]
    else:
        out[0] = 1.0
        out[2] = -f32(f32(out[0] * v[0]) + f32(out[1] * v[1])) / v[2]
    n2 = np.sqrt(float(out[0]) ** 2 + float(out[1]) ** 2 + float(out[2]) ** 2)     # cv::norm: double
    return np.array([f32(float(c) / n2) for c in out], np.float32)


def get_target_diameters(cam, size, targets, d_nodes, normals):
    """get_target_diameters (psp_process.cpp:114-165): image diameter = mean over 4 points of a
    circle in the local tangent plane (normal of the nearest model node)."""
    diams = np.zeros(len(targets), np.float32)
    live = [i for i, t in enumerate(targets) if t.diameter != 0 and contains(size, t.uv)]
    if not live:
        return diams
    near = engine.nearest_nodes(d_nodes, np.stack([targets[i].xyz for i in live]).astype(np.float64)).cpu().numpy()
    normals = np.asarray(normals, dtype=np.float32).reshape(-1, 3)
    pts = []
    for j, i in enumerate(live):
        t = targets[i]
        nrm = normals[near[j]]
        a = get_perpendicular(nrm)
        b = np.array([f32(f32(a[1] * nrm[2]) - f32(a[2] * nrm[1])),
                      f32(f32(a[2] * nrm[0]) - f32(a[0] * nrm[2])),
                      f32(f32(a[0] * nrm[1]) - f32(a[1] * nrm[0]))], np.float32)
        theta = f32(0.0)
        for _ in range(4):
            ca = 0.5 * float(t.diameter) * float(f32(np.cos(theta)))      # double scalar
            sb = 0.5 * float(t.diameter) * float(f32(np.sin(theta)))
            pa = np.array([f32(float(c) * ca) for c in a], np.float32)    # Point3f * double -> float
            pb = np.array([f32(float(c) * sb) for c in b], np.float32)
            pts.append(((t.xyz + pa).astype(np.float32) + pb).astype(np.float32))
            theta = f32(float(theta) + 2 * PI / 4)
    uv = engine.project_points(cam, np.stack(pts))
    for j, i in enumerate(live):
        t = targets[i]
        acc = f32(0.0)
        for k in range(4):
            d = (uv[4 * j + k] - t.uv).astype(np.float32)
            acc = f32(float(acc) + 2.0 * math.sqrt(float(d[0]) * float(d[0]) + float(d[1]) * float(d[1])))
        diams[i] = f32(float(acc) / 4.0)
    return diams


def cluster_points(targets, bound_pts=4):
    """upsp::cluster_points (cpp/lib/patches.ipp:239-276): breadth-first grouping of targets
    whose image distance is <= bound_pts + mean diameter; order of discovery preserved."""
    clusters = []
    pts = list(range(len(targets)))
    while pts:
        cl = [targets[pts.pop(0)]]
        head = 0
        while head < len(cl):
            ref = cl[head]
            head += 1
            rest = []
            for i in pts:
                t = targets[i]
                d = (ref.uv - t.uv).astype(np.float32)
                dist = math.sqrt(float(d[0]) * float(d[0]) + float(d[1]) * float(d[1]))
                lim = float(f32(bound_pts)) + 0.5 * float(f32(ref.diameter + t.diameter))
                if dist <= lim:
                    cl.append(t)
                else:
                    rest.append(i)
            pts = rest
        clusters.append(cl)
    return clusters


def _target_box(t):
    """get_target_boundary(targ, t_min, t_max) (patches.ipp:279-285)."""
    h = 0.5 * float(t.diameter)
    return (int(math.floor(float(t.uv[0]) - h)), int(math.floor(float(t.uv[1]) - h)),
            int(math.ceil(float(t.uv[0]) + h)), int(math.ceil(float(t.uv[1]) + h)))


def get_target_boundary(t, bound_pts=2, buffer=0):
    """Single-target patch (patches.ipp:288-327): interior = bounding box of the disc,
    boundary = frame of thickness bound_pts at distance `buffer` around it.  x outer, y inner."""
    x0, y0, x1, y1 = _target_box(t)
    internal = [(x, y) for x in range(x0, x1 + 1) for y in range(y0, y1 + 1)]
    bounds = []
    for x in range(x0 - bound_pts - buffer, x1 + bound_pts + buffer + 1):
        for y in range(y0 - bound_pts - buffer, y1 + bound_pts + buffer + 1):
            if x < x0 - buffer or x > x1 + buffer or y < y0 - buffer or y > y1 + buffer:
                bounds.append((x, y))
    return internal, bounds


def get_cluster_boundary(targets, bound_pts=2, buffer=0):
    """Multi-target patch (patches.ipp:330-485): union of the target boxes on a local mesh,
    filled between the extreme marked cells of every column, then of every row; boundary =
    cells whose (bound+buffer) window touches the interior while their buffer window does not."""
    boxes = [_target_box(t) for t in targets]
    tx0 = min(b[0] for b in boxes) - (bound_pts + buffer)
    ty0 = min(b[1] for b in boxes) - (bound_pts + buffer)
    tx1 = max(max(b[2] for b in boxes), 0) + (bound_pts + buffer)    # t_max starts at (0,0)
    ty1 = max(max(b[3] for b in boxes), 0) + (bound_pts + buffer)
    dx, dy = tx1 - tx0 + 1, ty1 - ty0 + 1
    cl = np.zeros((dx, dy), np.int32)
    for b in boxes:
        cl[b[0] - tx0:b[2] - tx0 + 1, b[1] - ty0:b[3] - ty0 + 1] = 2
    for x in range(dx):
        ys = np.nonzero(cl[x] == 2)[0]
        if ys.size:
            cl[x, ys[0]:ys[-1] + 1] = 2
    for y in range(dy):
        xs = np.nonzero(cl[:, y] == 2)[0]
        if xs.size:
            cl[xs[0]:xs[-1] + 1, y] = 2
    internal, bounds = [], []
    is2 = cl == 2           # the 1-marks written while scanning never change an `== 2` test
    w = bound_pts + buffer
    for x in range(dx):
        ax0, ax1 = max(x - w, 0), min(x + w, dx - 1)
        bx0, bx1 = max(x - buffer, 0), min(x + buffer, dx - 1)
        for y in range(dy):
            if is2[x, y]:
                internal.append((x + tx0, y + ty0))
                continue
            ay0, ay1 = max(y - w, 0), min(y + w, dy - 1)
            if bound_pts > 0 and buffer > 0:
                by0, by1 = max(y - buffer, 0), min(y + buffer, dy - 1)
                if not is2[bx0:bx1 + 1, by0:by1 + 1].any() and is2[ax0:ax1 + 1, ay0:ay1 + 1].any():
                    bounds.append((x + tx0, y + ty0))
                continue
            if bound_pts > 0 and is2[ax0:ax1 + 1, ay0:ay1 + 1].any():
                bounds.append((x + tx0, y + ty0))
    return internal, bounds


def patch_clusters(clusters, size, boundary_thickness=2, buffer_thickness=1):
    """PatchClusters ctor (patches.ipp:14-54): pixel lists per cluster, clipped to the frame.
    Returns the list of dict(bx, by, ix, iy) the frame pipeline takes."""
    out = []
    for cl in clusters:
        if len(cl) > 1:
            internal, bounds = get_cluster_boundary(cl, boundary_thickness, buffer_thickness)
        else:
            internal, bounds = get_target_boundary(cl[0], boundary_thickness, buffer_thickness)
        inside = lambda p: 0 <= p[0] < size[0] and 0 <= p[1] < size[1]
        internal = [p for p in internal if inside(p)]
        bounds = [p for p in bounds if inside(p)]
        out.append(dict(ix=np.array([p[0] for p in internal], np.int32),
                        iy=np.array([p[1] for p in internal], np.int32),
                        bx=np.array([p[0] for p in bounds], np.int32),
                        by=np.array([p[1] for p in bounds], np.int32)))
    return out


def threshold_bounds(patches, ref, thresh, offset=2):
    """PatchClusters::threshold_bounds (patches.ipp:57-94): drops every boundary pixel whose
    (2*offset+1)^2 neighbourhood (clipped to the frame) holds a value below thresh."""
    ref = np.asarray(ref)
    rows, cols = ref.shape
    for p in patches:
        keep = []
        for k in range(p["bx"].size):
            x, y = int(p["bx"][k]), int(p["by"][k])
            x0, y0 = max(0, x - offset), max(0, y - offset)
            x1, y1 = min(cols - 1, x + offset), min(rows - 1, y + offset)
            keep.append(not (float(ref[y0:y1 + 1, x0:x1 + 1].min()) < float(thresh)))
        keep = np.array(keep, dtype=bool) if keep else np.zeros(0, bool)
        p["bx"], p["by"] = p["bx"][keep], p["by"][keep]
    return patches


def intensity_histc(img, depth=12, bins=-1):
    """upsp::intensity_histc (cpp/lib/image_processing.ipp:10-50) for u16 / u8 images."""
    img = np.asarray(img)
    im_depth = 8 if img.dtype == np.uint8 else 16
    depth = min(int(depth), im_depth)
    max_value = 1 << depth
    if bins == -1:
        bins = max_value
    bin_sz = int(math.ceil(max_value // bins))       # integer division first, like the reference
    v = img.reshape(-1).astype(np.int64)
    v = v[v < max_value]
    counts = np.bincount(v // bin_sz, minlength=bins)[:bins].astype(np.int64)
    edges = (np.arange(bins + 1) * bin_sz).astype(np.int64)
    return edges, counts


def find_peaks(data, separation=0):
    """upsp::find_peaks (cpp/utils/clustering.ipp:9-60), including its early exit: a peak closer
    than `separation` to the previous one replaces it if higher and ENDS the scan (`break`)."""
    peaks = []
    n = len(data)
    if n < 3:
        return peaks
    plateau, begin = False, 0
    for i in range(1, n - 1):
        d, dm, dp = data[i], data[i - 1], data[i + 1]
        if math.isinf(d) or (d > dm and d > dp):
            if peaks and (i - peaks[-1]) < separation:
                if data[peaks[-1]] < d:
                    peaks[-1] = i
                break
            peaks.append(i)
        elif d > dm and d == dp:
            plateau, begin = True, i
        elif plateau:
            if d < dp:
                plateau = False
            elif d > dp:
                plateau = False
                pi = (i + begin) // 2
                if peaks and (pi - peaks[-1]) < separation:
                    if data[peaks[-1]] < data[pi]:
                        peaks[-1] = pi
                    break
                peaks.append(pi)
    return peaks


def first_min_threshold(counts, separation=1):
    """upsp::first_min_threshold (clustering.ipp:62-96): first minimum after the first maximum."""
    counts = [int(c) for c in counts]
    maxp = find_peaks(counts, separation)
    if not maxp:
        return 0
    inv = [math.inf if c == 0 else 1.0 / c for c in counts]
    minp = find_peaks(inv, separation)
    for p in minp:
        if p > maxp[0]:
            return p
    return 0


def initialize_image_patches(bvh, cam, size, target_file, first_frame, d_nodes, normals,
                             oblique_angle=70.0, bit_depth=12, bound_pts=2, buffer_pts=1,
                             target_diam_sf=1.2):
    """InitializeImagePatches for one camera (psp_process.cpp:2092-2164).  first_frame: raw u16
    [H,W] array (frame 1).  Returns (patch tables, visible targets, threshold)."""
    targs = read_psp_target_file(target_file) + read_psp_target_file(target_file, "*Fiducials")
    vis = get_targets(bvh, cam, size, targs, d_nodes, normals, target_oblique_threshold(oblique_angle))
    map_points_to_image(cam, vis)
    diams = get_target_diameters(cam, size, vis, d_nodes, normals)
    for t, d in zip(vis, diams):
        t.diameter = f32(d * f32(target_diam_sf))
    clusters = cluster_points(vis, bound_pts + buffer_pts)
    frame = np.asarray(first_frame)
    edges, counts = intensity_histc(frame, bit_depth, 256)
    thresh = int(edges[first_min_threshold(counts, 5)]) + 5
    patches = patch_clusters(clusters, size, bound_pts, buffer_pts)
    threshold_bounds(patches, frame, thresh, 2)
    return patches, vis, thresh
