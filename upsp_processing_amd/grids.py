"""Grid loaders and the structured-surface model feeding the hot path (last row).

* PLOT3D unformatted grid files, single / multi zone, little / big endian, single / double
  precision, with / without IBLANK (upsp::read_plot3d_grid_file / write_plot3d_grid_file,
  cpp/lib/plot3d.cpp:103-329) and scalar function files (read_plot3d_scalar_function_file,
  cpp/lib/plot3d.cpp:41-101);
* `P3DModel`: upsp::P3DModel_<float> reduced to what psp_process phase 1 uses
  (cpp/lib/P3DModel.ipp): node numbering, identifyOverlap (:893-1127), calcNormals (:1357-1654),
  extract_tris (:233-320), adjust_solution (:143-157), is_superceded (:700-703).

Host-side geometry preprocessing, run once per model; the arrays it produces (soup, tri->node
ids, nodes, normals, overlap source map) go to the GPU engine.
"""
import struct

import numpy as np

f32 = np.float32


# ----------------------------------------------------------------------------- PLOT3D --
def read_plot3d_grid(path, dtype=np.float32):
    """Returns dict(zones=[(j,k,l)...], x, y, z) with x/y/z concatenated over zones (dtype)."""
    with open(path, "rb") as f:
        buf = f.read()
    if len(buf) < 4:
        raise ValueError("Could not read plot3d file %s" % path)
    b0, b3 = buf[0], buf[3]
    if b0 == 0 and b3 != 0:
        end = ">"
    elif b0 == 0 and b3 == 0:
        raise ValueError("Unable to identify plot3d endianness in file %s" % path)
    else:
        end = "<"
    i32 = end + "i4"
    off = 0

    def take(fmt, n):
        nonlocal off
        a = np.frombuffer(buf, fmt, n, off)
        off += a.nbytes
        return a

    first = int(take(i32, 1)[0])
    if first == 4:                                  # multi-zone: record {nzones}
        zones = int(take(i32, 1)[0])
        take(i32, 1)
        take(i32, 1)                                # leading marker of the size record
    else:
        if first != 12:
            raise ValueError("Unexpected plot3d file format in file %s" % path)
        zones = 1
    sizes = take(i32, 3 * zones).reshape(zones, 3).astype(np.int64)
    take(i32, 1)
    xs, ys, zs = [], [], []
    data_fmt, has_iblank = None, False
    for zi in range(zones):
        n = int(np.prod(sizes[zi]))
        rec = int(take(i32, 1)[0])
        if zi == 0:                                 # precision / IBLANK from the first record
            if rec == 12 * n:
                data_fmt, has_iblank = end + "f4", False
            elif rec == 16 * n:
                data_fmt, has_iblank = end + "f4", True
            elif rec == 24 * n:
                data_fmt, has_iblank = end + "f8", False
            elif rec == 28 * n:
                data_fmt, has_iblank = end + "f8", True
            else:
                raise ValueError("Unrecognized data type in plot3d file %s" % path)
        xs.append(take(data_fmt, n).astype(dtype))
        ys.append(take(data_fmt, n).astype(dtype))
        zs.append(take(data_fmt, n).astype(dtype))
        if has_iblank:
            off += 4 * n
        take(i32, 1)
    return dict(zones=[tuple(int(v) for v in s) for s in sizes],
                x=np.concatenate(xs), y=np.concatenate(ys), z=np.concatenate(zs))


def write_plot3d_grid(path, grid):
    """write_plot3d_grid_file: little endian, no IBLANK, precision of grid['x'].dtype; a single
    zone is written in the single-grid form (no zone-count record)."""
    zones = grid["zones"]
    dt = np.dtype(grid["x"].dtype).newbyteorder("<")
    with open(path, "wb") as f:
        if len(zones) != 1:
            f.write(struct.pack("<iii", 4, len(zones), 4))
        rec = 12 * len(zones)
        f.write(struct.pack("<i", rec))
        for z in zones:
            f.write(struct.pack("<iii", *z))
        f.write(struct.pack("<i", rec))
        idx = 0
        for z in zones:
            n = z[0] * z[1] * z[2]
            rec = dt.itemsize * n * 3
            f.write(struct.pack("<i", rec))
            for a in (grid["x"], grid["y"], grid["z"]):
                f.write(np.ascontiguousarray(a[idx:idx + n], dtype=dt).tobytes())
            f.write(struct.pack("<i", rec))
            idx += n


def read_plot3d_scalar_function_file(path, record_seps=-1):
    """One f32 scalar per grid point, all zones (native little endian).  record_seps: +1 = with
    FORTRAN record separators, 0 = without, -1 = try with, then without.
    Like the reference, the scalar record itself is read without looking at separators
    (plot3d.cpp:69): in a file WITH separators the leading marker lands in scalar 0 and the
    values are shifted by one; the reference's test only checks the count."""
    with open(path, "rb") as f:
        buf = f.read()

    def parse(seps):
        off = 0

        def rec(fmt, n, with_seps):
            nonlocal off
            nbytes = n * np.dtype(fmt).itemsize
            if with_seps:
                if off + 4 > len(buf) or struct.unpack_from("<i", buf, off)[0] != nbytes:
                    raise ValueError("record separator")
                off += 4
            if off + nbytes > len(buf):
                raise ValueError("short record")
            a = np.frombuffer(buf, fmt, n, off)
            off += nbytes
            if with_seps:
                if off + 4 > len(buf) or struct.unpack_from("<i", buf, off)[0] != nbytes:
                    raise ValueError("record separator")
                off += 4
            return a
        nz = int(rec("<i4", 1, seps)[0])
        if nz < 0 or nz > 1 << 20:
            raise ValueError("number of zones")
        sz = rec("<i4", 4 * nz, seps).reshape(nz, 4)
        total = int((sz[:, 0].astype(np.int64) * sz[:, 1] * sz[:, 2]).sum())
        # the reference reads the scalars without looking at separators (plot3d.cpp:69)
        return rec("<f4", total, False).copy()

    errs = []
    if record_seps in (1, -1):
        try:
            return parse(True)
        except ValueError as e:
            errs.append("assuming record separators: %s" % e)
    if record_seps in (0, -1):
        try:
            return parse(False)
        except ValueError as e:
            errs.append("assuming no record separators: %s" % e)
    raise ValueError("Failed to parse Plot3D function file %r: %s" % (path, "; ".join(errs)))


# --------------------------------------------------------------------------- P3DModel --
class P3DModel:
    """Structured surface model: zones of (j, k, 1) nodes, node index = zone offset + k*J + j."""

    def __init__(self, grid, tol=0.0):
        self.zones = [tuple(z) for z in grid["zones"]]
        for z in self.zones:
            if z[2] != 1:
                raise ValueError("P3DModel needs surface zones (l == 1)")     # P3DModel.ipp:73-76
        self.x = np.ascontiguousarray(grid["x"], dtype=np.float32)
        self.y = np.ascontiguousarray(grid["y"], dtype=np.float32)
        self.z = np.ascontiguousarray(grid["z"], dtype=np.float32)
        self.start = np.concatenate([[0], np.cumsum([z[0] * z[1] for z in self.zones])]).astype(np.int64)
        self.nnodes = int(self.start[-1])
        assert self.x.size == self.nnodes
        self.tol = float(tol)
        self.overlap = self._identify_overlap(self.tol)
        self.normals = self._calc_normals()

    @classmethod
    def from_file(cls, path, tol=0.0):
        return cls(read_plot3d_grid(path, np.float32), tol)

    # -- sizes ---------------------------------------------------------------------------
    def size(self):
        return self.nnodes

    def number_of_faces(self):
        return int(sum((z[0] - 1) * (z[1] - 1) for z in self.zones))

    def nodes(self):
        return np.stack([self.x, self.y, self.z], axis=1)

    def zone_of(self, nidx):
        return int(np.searchsorted(self.start, nidx, side="right") - 1)

    # -- overlap (identifyOverlap, P3DModel.ipp:893-1127) ----------------------------------
    def _boundary_nodes(self):
        out = []
        for zi, (J, K, _) in enumerate(self.zones):
            idx = np.arange(J * K)
            r, c = idx // J, idx % J
            m = (r == 0) | (r == K - 1) | (c == 0) | (c == J - 1)
            out.append(self.start[zi] + idx[m])
        return np.concatenate(out) if out else np.zeros(0, np.int64)

    def _identify_overlap(self, tol):
        from scipy.spatial import cKDTree
        tol = float(f32(tol))
        tol = tol if tol > float(f32(1e-12)) else float(f32(1e-12))
        b = self._boundary_nodes()
        if b.size == 0:
            return {}
        P = np.stack([self.x[b], self.y[b], self.z[b]], axis=1).astype(np.float64)
        tree = cKDTree(P)
        # candidates with a slightly larger radius; the decision is the kd-tree's own:
        # sum of squared double differences (x, y, z order) <= range^2 (pspKdtree.c:226-241)
        pairs = tree.query_pairs(tol * (1 + 1e-9) + 1e-300, output_type="ndarray")
        zone = np.searchsorted(self.start, b, side="right") - 1
        J = np.array([z[0] for z in self.zones])[zone]
        K = np.array([z[1] for z in self.zones])[zone]
        loc = b - self.start[zone]
        kk, jj = loc // J, loc % J
        over = {}
        for a, c in pairs:
            d = P[a] - P[c]
            if (0.0 + d[0] * d[0] + d[1] * d[1]) + d[2] * d[2] > tol * tol:
                continue
            if zone[a] == zone[c]:           # same zone: only where the zone wraps onto itself
                wrapped = False
                if jj[a] == jj[c]:
                    wrapped = {int(kk[a]), int(kk[c])} == {0, int(K[a]) - 1} and K[a] > 1
                elif kk[a] == kk[c]:
                    wrapped = {int(jj[a]), int(jj[c])} == {0, int(J[a]) - 1} and J[a] > 1
                if not wrapped:
                    continue
            na, nc = int(b[a]), int(b[c])
            over.setdefault(na, set()).add(nc)
            over.setdefault(nc, set()).add(na)
        return {k: sorted(v) for k, v in sorted(over.items())}

    def is_overlapping(self, nidx):
        return int(nidx) in self.overlap

    def get_low_nidx(self, nidx):
        o = self.overlap.get(int(nidx))
        return o[0] if o and o[0] < nidx else int(nidx)

    def is_superceded(self, nidx):
        return self.get_low_nidx(nidx) != int(nidx)

    def overlap_source(self):
        """src[n] = node whose value node n holds after adjust_solution (P3DModel.ipp:143-157):
        the loop runs over the overlap map in ascending key order and copies sol[curr] into every
        higher-numbered overlapping node, so copies chain through already-overwritten entries."""
        src = np.arange(self.nnodes, dtype=np.int32)
        for curr, others in self.overlap.items():
            for alt in others:
                if curr < alt:
                    src[alt] = src[curr]
        return src

    def adjust_solution(self, sol):
        sol = np.asarray(sol)
        return sol[self.overlap_source()]

    # -- normals (calcNormals, P3DModel.ipp:1357-1654) --------------------------------------
    def _zone_face_normals(self, zi):
        """Unit normals of the 4 faces around every node of zone zi, float arithmetic, zero where
        the face does not exist.  Returns [4][K*J, 3] in the order LL, LR, UR, UL."""
        J, K, _ = self.zones[zi]
        s = int(self.start[zi])
        X = np.stack([self.x[s:s + J * K], self.y[s:s + J * K], self.z[s:s + J * K]], axis=1)
        X = X.reshape(K, J, 3)
        r, c = np.meshgrid(np.arange(K), np.arange(J), indexing="ij")
        cm, cp = np.maximum(c - 1, 0), np.minimum(c + 1, J - 1)
        rm, rp = np.maximum(r - 1, 0), np.minimum(r + 1, K - 1)
        me = X[r, c]
        cases = (  # (v0, v2, valid): faceNormal(v0, v1 = node, v2)
            (X[rm, c], X[r, cm], (cm != c) & (rm != r)),      # LL: v0 = rm, v2 = cm
            (X[r, cp], X[rm, c], (cp != c) & (rm != r)),      # LR: v0 = cp, v2 = rm
            (X[rp, c], X[r, cp], (cp != c) & (rp != r)),      # UR: v0 = rp, v2 = cp
            (X[r, cm], X[rp, c], (cm != c) & (rp != r)),      # UL: v0 = cm, v2 = rp
        )
        out = []
        for v0, v2, valid in cases:
            u = (v2 - me).astype(np.float32)
            v = (v0 - me).astype(np.float32)
            nx = (u[..., 1] * v[..., 2]).astype(np.float32) - (v[..., 1] * u[..., 2]).astype(np.float32)
            ny = (v[..., 0] * u[..., 2]).astype(np.float32) - (u[..., 0] * v[..., 2]).astype(np.float32)
            nz = (u[..., 0] * v[..., 1]).astype(np.float32) - (v[..., 0] * u[..., 1]).astype(np.float32)
            n = np.stack([nx, ny, nz], axis=-1).astype(np.float32)
            mag = np.sqrt((n.astype(np.float64) ** 2).sum(-1)).astype(np.float32)       # cv::norm -> FP
            safe = np.where(mag == 0, f32(1), mag)
            n = np.where((mag == 0)[..., None], n, (n / safe[..., None]).astype(np.float32))
            n = np.where(valid[..., None], n, f32(0)).astype(np.float32)
            out.append(n.reshape(K * J, 3))
        return out

    def _calc_normals(self):
        faces = [self._zone_face_normals(zi) for zi in range(len(self.zones))]
        own = np.zeros((self.nnodes, 3), np.float32)
        for zi, fz in enumerate(faces):
            s, e = int(self.start[zi]), int(self.start[zi + 1])
            acc = np.zeros((e - s, 3), np.float32)
            for fn in fz:                                   # LL, LR, UR, UL in this order
                acc = (acc + fn).astype(np.float32)
            own[s:e] = acc
        total = own.copy()
        # nodes with overlaps add the faces of every overlapping node, ascending, face by face
        for n, others in self.overlap.items():
            acc = own[n].copy()
            for o in others:
                zi = self.zone_of(o)
                lo = o - int(self.start[zi])
                for fn in faces[zi]:
                    acc = (acc + fn[lo]).astype(np.float32)
            total[n] = acc
        mag = np.sqrt((total.astype(np.float64) ** 2).sum(1)).astype(np.float32)
        safe = np.where(mag == 0, f32(1), mag)
        return np.where((mag == 0)[:, None], total, (total / safe[:, None]).astype(np.float32)).astype(np.float32)

    # -- triangles (extract_tris, P3DModel.ipp:233-320) -----------------------------------
    def extract_tris(self):
        """Two triangles per quad, (i0,i1,i2) and (i2,i3,i0).  Returns (soup f32[9T], triNodes i32[3T])."""
        tn = []
        for zi, (J, K, _) in enumerate(self.zones):
            q = np.arange((J - 1) * (K - 1))
            klo, jlo = q // (J - 1), q % (J - 1)
            base = int(self.start[zi])
            i0 = base + klo * J + jlo
            i1 = i0 + 1
            i2 = base + (klo + 1) * J + jlo + 1
            i3 = i2 - 1
            tn.append(np.stack([i0, i1, i2, i2, i3, i0], axis=1).reshape(-1, 3))
        tn = np.concatenate(tn).astype(np.int32) if tn else np.zeros((0, 3), np.int32)
        soup = self.nodes()[tn].reshape(-1).astype(np.float32)
        return soup, tn.reshape(-1)
