"""Deterministic synthetic inputs for tests and bench.

Everything here is host-side input generation (numpy) or device-side frame synthesis
(torch elementwise ops); none of it is on the measured path.
"""
import math

import numpy as np


def uv_sphere(nlat, nlon, radius=1.0, scale=(1.0, 1.0, 1.0), offset=(0.0, 0.0, 0.0)):
    """UV sphere with fan caps: (nlat-1)*nlon + 2 vertices, 2*nlon*(nlat-1) triangles,
    counter-clockwise seen from outside.  Returns (verts f32 [V,3], tris int32 [T,3])."""
    th = np.pi * np.arange(1, nlat) / nlat                      # polar angle of the rings
    ph = 2 * np.pi * np.arange(nlon) / nlon
    st, ct = np.sin(th)[:, None], np.cos(th)[:, None]
    ring = np.stack([st * np.cos(ph)[None], st * np.sin(ph)[None], np.broadcast_to(ct, (nlat - 1, nlon))], -1)
    verts = np.concatenate([[[0.0, 0.0, 1.0]], ring.reshape(-1, 3), [[0.0, 0.0, -1.0]]])
    verts = verts * radius * np.asarray(scale)[None] + np.asarray(offset)[None]
    north, south = 0, verts.shape[0] - 1

    def vid(i, j):  # ring i (0..nlat-2), longitude j
        return 1 + i * nlon + (j % nlon)

    j = np.arange(nlon)
    tris = [np.stack([np.full(nlon, north), vid(0, j), vid(0, j + 1)], 1)]
    for i in range(nlat - 2):
        a, b, c, d = vid(i, j), vid(i + 1, j), vid(i + 1, j + 1), vid(i, j + 1)
        tris.append(np.stack([a, b, c], 1))
        tris.append(np.stack([a, c, d], 1))
    tris.append(np.stack([np.full(nlon, south), vid(nlat - 2, j + 1), vid(nlat - 2, j)], 1))
    return verts.astype(np.float32), np.concatenate(tris).astype(np.int32)


def cube_sphere(m, radius=1.0, scale=(1.0, 1.0, 1.0), offset=(0.0, 0.0, 0.0)):
    """Sphere from a subdivided cube (m x m quads per face, 12 m^2 triangles, 6 m^2 + 2
    vertices, valence <= 6 everywhere), counter-clockwise seen from outside."""
    g = np.arange(m + 1)
    a, b = np.meshgrid(g, g, indexing="ij")
    a, b = a.ravel(), b.ravel()
    zero, full = np.zeros_like(a), np.full_like(a, m)
    # (lattice point, (du, dv)) with du x dv pointing outward
    faces = [(np.stack([full, a, b], 1)), (np.stack([zero, b, a], 1)),
             (np.stack([b, full, a], 1)), (np.stack([a, zero, b], 1)),
             (np.stack([a, b, full], 1)), (np.stack([b, a, zero], 1))]
    pts = np.concatenate(faces)
    key = (pts[:, 0] * (m + 1) + pts[:, 1]) * (m + 1) + pts[:, 2]
    uniq, first, inv = np.unique(key, return_index=True, return_inverse=True)
    lattice = pts[first].astype(np.float64) / m * 2.0 - 1.0
    verts = lattice / np.linalg.norm(lattice, axis=1, keepdims=True)
    verts = verts * radius * np.asarray(scale)[None] + np.asarray(offset)[None]
    i, j = np.meshgrid(np.arange(m), np.arange(m), indexing="ij")
    i, j = i.ravel(), j.ravel()
    tris = []
    for f in range(6):
        base = f * (m + 1) ** 2
        p00 = inv[base + i * (m + 1) + j]
        p10 = inv[base + (i + 1) * (m + 1) + j]
        p11 = inv[base + (i + 1) * (m + 1) + j + 1]
        p01 = inv[base + i * (m + 1) + j + 1]
        tris.append(np.stack([p00, p10, p11], 1))
        tris.append(np.stack([p00, p11, p01], 1))
    return verts.astype(np.float32), np.concatenate(tris).astype(np.int32)


def merge_meshes(parts):
    verts, tris, base = [], [], 0
    for v, t in parts:
        verts.append(v)
        tris.append(t + base)
        base += v.shape[0]
    return np.concatenate(verts).astype(np.float32), np.concatenate(tris).astype(np.int32)


def tunnel_model(nlat=400, nlon=1000, blat=160, blon=320):
    """'Wind-tunnel model' from UV spheres (polar fans: two vertices of valence nlon):
    sphere stretched x6 along x plus two offset booster ellipsoids that occlude part of
    the body.  Defaults: 1 001 520 triangles, 500 766 nodes."""
    body = uv_sphere(nlat, nlon, 1.0, scale=(6.0, 1.0, 1.0))
    b1 = uv_sphere(blat, blon, 0.45, scale=(5.0, 1.0, 1.0), offset=(-1.0, 1.25, 0.35))
    b2 = uv_sphere(blat, blon, 0.45, scale=(5.0, 1.0, 1.0), offset=(-1.0, -1.25, 0.35))
    return merge_meshes([body, b1, b2])


def tunnel_model_quad(m=258, mb=92):
    """'Wind-tunnel model' from cube spheres (structured-grid-like, valence <= 6): the
    same body + two boosters layout.  Defaults: 12*(258^2 + 2*92^2) = 1 001 904 triangles,
    500 958 nodes."""
    body = cube_sphere(m, 1.0, scale=(6.0, 1.0, 1.0))
    b1 = cube_sphere(mb, 0.45, scale=(5.0, 1.0, 1.0), offset=(-1.0, 1.25, 0.35))
    b2 = cube_sphere(mb, 0.45, scale=(5.0, 1.0, 1.0), offset=(-1.0, -1.25, 0.35))
    return merge_meshes([body, b1, b2])


def soup(verts, tris):
    """extract_tris layout (cpp/lib/TriModel.ipp:261-299): 9 floats per triangle + tri->node ids."""
    return verts[tris].reshape(-1).astype(np.float32), tris.reshape(-1).astype(np.int32)


def node_normals(verts, tris):
    """TriModel_::calcNormals (cpp/lib/TriModel.ipp:1429-1506): normalised sum of unit face
    normals (n2-n1)x(n0-n1); zero stays zero.  float32 arithmetic."""
    v = verts.astype(np.float32)
    n0, n1, n2 = v[tris[:, 0]], v[tris[:, 1]], v[tris[:, 2]]
    fn = np.cross(n2 - n1, n0 - n1).astype(np.float32)
    mag = np.linalg.norm(fn.astype(np.float64), axis=1).astype(np.float32)
    fn = np.where(mag[:, None] == 0, fn, fn / np.where(mag == 0, 1, mag)[:, None]).astype(np.float32)
    acc = np.zeros_like(v)
    for k in range(3):
        np.add.at(acc, tris[:, k], fn)
    mag = np.linalg.norm(acc.astype(np.float64), axis=1).astype(np.float32)
    return np.where(mag[:, None] == 0, acc, acc / np.where(mag == 0, 1, mag)[:, None]).astype(np.float32)


def pinhole_camera(width, height, center=(0.0, 0.0, 4.0), half_extent=1.0, fill=0.7, k1=0.0,
                   azimuth_deg=0.0):
    """Camera at `center` (rotated about the model x axis by azimuth) looking at the origin.
    Returns dict(K, dist, R, t, width, height) with OpenCV conventions (x right, y down,
    z forward); R is model->camera."""
    c = np.asarray(center, dtype=np.float64)
    a = math.radians(azimuth_deg)
    rot = np.array([[1, 0, 0], [0, math.cos(a), -math.sin(a)], [0, math.sin(a), math.cos(a)]])
    c = rot @ c
    zc = -c / np.linalg.norm(c)                   # forward
    up = rot @ np.array([0.0, 1.0, 0.0])
    xc = np.cross(-up, zc)                        # y axis points down in the image
    xc /= np.linalg.norm(xc)
    yc = np.cross(zc, xc)
    R = np.stack([xc, yc, zc])
    t = -R @ c
    dist = np.linalg.norm(c)
    f = 0.5 * fill * width * dist / half_extent
    K = np.array([[f, 0, width / 2.0], [0, f, height / 2.0], [0, 0, 1.0]])
    return dict(K=K, dist=np.array([k1, 0, 0, 0, 0.0]), R=R, t=t, width=width, height=height)


def frame_params(nframes, seed=20240607):
    """Per-frame affine jitter A_f = I + small (<=0.5 px translation, <=1e-3 shear)."""
    rng = np.random.default_rng(seed)
    A = np.zeros((nframes, 2, 3))
    A[:, 0, 0] = 1 + rng.uniform(-1e-3, 1e-3, nframes)
    A[:, 1, 1] = 1 + rng.uniform(-1e-3, 1e-3, nframes)
    A[:, 0, 1] = rng.uniform(-1e-3, 1e-3, nframes)
    A[:, 1, 0] = rng.uniform(-1e-3, 1e-3, nframes)
    A[:, :, 2] = rng.uniform(-0.5, 0.5, (nframes, 2))
    return A


def synth_frames_numpy(nframes, height, width, first=0, seed=20240607, noise=8.0, hot=False):
    """I(x,y,f) = 1800 + 900 sin(2pi 3x'/W) cos(2pi 2y'/H) + 50 sin(2pi f/64) + noise,
    12-bit u16.  Small sizes only (host)."""
    A = frame_params(first + nframes, seed)[first:]
    rng = np.random.default_rng(seed + 1 + first)
    y, x = np.mgrid[0:height, 0:width].astype(np.float64)
    out = np.zeros((nframes, height, width), np.uint16)
    for i in range(nframes):
        xp = A[i, 0, 0] * x + A[i, 0, 1] * y + A[i, 0, 2]
        yp = A[i, 1, 0] * x + A[i, 1, 1] * y + A[i, 1, 2]
        img = (1800 + 900 * np.sin(2 * np.pi * 3 * xp / width) * np.cos(2 * np.pi * 2 * yp / height)
               + 50 * np.sin(2 * np.pi * (first + i) / 64))
        img += rng.normal(0, noise, img.shape)
        img = np.clip(np.rint(img), 0, 4095)
        if hot and (first + i) % 7 == 3:
            for _ in range(int(rng.integers(1, 4))):
                img[int(rng.integers(0, height)), int(rng.integers(0, width))] = 4095
        out[i] = img.astype(np.uint16)
    return out


def hot_pixel_positions(frame, height, width, seed=20240607, active=None):
    """<= 3 injected hot pixels (value 4095) in 1 % of the frames.  Deterministic in
    (seed, global frame index).  Returns a list of flat pixel positions (empty for 99 % of the frames);
    with `active` (flat indices of pixels some node reads) the first one lands on such a pixel, so that
    the repair reaches the time series."""
    if frame % 100 != 17:
        return []
    rng = np.random.default_rng(seed + 7919 * (frame + 1))
    n = int(rng.integers(1, 4))
    pos = [int(rng.integers(0, height * width)) for _ in range(n)]
    if active is not None and len(active):
        pos[0] = int(active[int(rng.integers(0, len(active)))])
    return pos


def scene_layout(pix, height, width, ndiscs=24):
    """Static content of the synthetic frames derived from a projection: the model
    silhouette (pixels some node reads, dilated by 2 px; everything else is background = 60) and 24
    fiducial discs (radius 3-5 px, intensity x 0.3) centred on evenly spaced visible nodes.
    pix: int32 [N] (numpy or tensor).  Returns dict(mask bool [H,W], scale f32 [H,W], active int64 [A])
    as numpy arrays."""
    pix = np.asarray(pix.cpu() if hasattr(pix, "cpu") else pix).reshape(-1)
    act = np.unique(pix[pix >= 0]).astype(np.int64)
    m = np.zeros((height, width), bool)
    m.flat[act] = True
    d = m.copy()
    for dy in range(-2, 3):                     # 5 x 5 dilation
        for dx in range(-2, 3):
            sh = np.zeros_like(m)
            ys, yd = (slice(max(dy, 0), height + min(dy, 0)), slice(max(-dy, 0), height + min(-dy, 0)))
            xs, xd = (slice(max(dx, 0), width + min(dx, 0)), slice(max(-dx, 0), width + min(-dx, 0)))
            sh[yd, xd] = m[ys, xs]
            d |= sh
    scale = np.ones((height, width), np.float32)
    if act.size and ndiscs > 0:
        yy, xx = np.mgrid[0:height, 0:width]
        for i in range(ndiscs):
            c = int(act[(i * act.size) // ndiscs + act.size // (2 * ndiscs)])
            cy, cx, r = c // width, c % width, 3 + i % 3
            scale[(yy - cy) ** 2 + (xx - cx) ** 2 <= r * r] = 0.3
    return dict(mask=d, scale=scale, active=act)


def synth_frames_torch(nframes, height, width, first=0, seed=20240607, noise=8.0, device="cuda",
                       out=None, layout=None, hot=False):
    """Same image model synthesised on the device (bench input; noise from torch's
    generator, so values differ from synth_frames_numpy).  layout = scene_layout(...): background 60
    outside the model, fiducial discs x 0.3; hot=True: hot_pixel_positions() (1 % of the frames)."""
    import torch
    A = torch.as_tensor(frame_params(first + nframes, seed)[first:], dtype=torch.float32, device=device)
    g = torch.Generator(device=device)
    g.manual_seed(seed + 1 + first)
    y = torch.arange(height, device=device, dtype=torch.float32)[:, None]
    x = torch.arange(width, device=device, dtype=torch.float32)[None, :]
    if out is None:
        out = torch.empty((nframes, height, width), dtype=torch.uint16, device=device)
    mask = scale = active = None
    if layout is not None:
        mask = torch.as_tensor(layout["mask"], device=device)
        scale = torch.as_tensor(layout["scale"], device=device)
        active = layout["active"]
    for i in range(nframes):
        xp = A[i, 0, 0] * x + A[i, 0, 1] * y + A[i, 0, 2]
        yp = A[i, 1, 0] * x + A[i, 1, 1] * y + A[i, 1, 2]
        img = (1800 + 900 * torch.sin(2 * math.pi * 3 * xp / width) * torch.cos(2 * math.pi * 2 * yp / height)
               + 50 * math.sin(2 * math.pi * (first + i) / 64))
        if layout is not None:
            img = torch.where(mask, img * scale, torch.full_like(img, 60.0))
        img = img + noise * torch.randn(img.shape, generator=g, device=device)
        out[i] = img.round().clamp_(0, 4095).to(torch.int32).to(torch.uint16)
        if hot:
            pos = hot_pixel_positions(first + i, height, width, seed, active)
            if pos:
                flat = out[i].view(torch.int16).reshape(-1)
                flat[torch.as_tensor(pos, device=device)] = 4095
    return out
