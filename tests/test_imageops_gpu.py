"""GPU parity of the image stages (register / patch / filter) vs the CPU oracle.

PARITY UNPINNED against the reference (arithmetic in OpenCV / Eigen, no reference
test).  Against the oracle's restatement the bars are:
  blur (gaussian, box)       bit-exact (same float / double operation order)
  warpAffine of a u16 frame  bit-exact for a given warp matrix (integer fixed-point
                             coordinates, 1/32-px weights)
  ECC warp matrix            max|dM| <= 1e-4 on the linear part, <= 2e-3 px on the
                             translation (the GPU folds the iteration
                             into one pass of double sums, the oracle keeps OpenCV's
                             float intermediates)
  patched pixels             bit-exact vs the oracle's float column-pivoted QR (reference arithmetic); the
                             oracle's (= reference's) float QR on raw pixel coordinates
                             is itself only good to ~5e-3
"""
import numpy as np
import pytest

from test_image_oracle import disc_cluster

pytestmark = pytest.mark.gpu


def test_blur_bitwise(gpu_lib, oracle):
    import torch
    from upsp_processing_amd import engine
    rng = np.random.default_rng(4)
    for shape in [(37, 53), (128, 96), (5, 7)]:
        img = (rng.normal(size=shape) * 500 + 1800).astype(np.float32)
        d = torch.as_tensor(img).cuda()
        for k in (1, 3, 5, 7, 9, 15):
            if k // 2 >= min(shape) * 2:
                continue
            assert np.array_equal(engine.blur(d, k).cpu().numpy().view(np.int32),
                                  oracle.blur(img, k).view(np.int32)), (shape, k)
        for k in (3, 5):
            assert np.array_equal(engine.blur(d, k, box=True).cpu().numpy().view(np.int32),
                                  oracle.blur(img, k, box=True).view(np.int32)), (shape, k)


def test_blur_u16_bitwise(gpu_lib, oracle):
    """upsp_blur_u16 (convertTo + GaussianBlur of u16 frames in one pass, the registration's pre-blur) against the
    oracle's GaussianBlur of the converted frame, bit for bit -- widths that are not multiples of four take the fused tile
    kernel (64 x 32 tile + halo in LDS): widths and heights around the tiles, images smaller than the kernel's halo,
    several frames in one call; other kernel sizes."""
    import torch
    from upsp_processing_amd import engine
    rng = np.random.default_rng(9)
    for shape in [(5, 5), (7, 61), (64, 62), (65, 63), (130, 124), (33, 125), (200, 249), (129, 300), (70, 1024 + 3)]:
        fr = rng.integers(0, 4096, size=(3,) + shape, dtype=np.uint16)
        fr[0, 0, :] = 4095
        fr[1, :, -1] = 0
        g = engine.blur_u16(torch.as_tensor(fr).cuda(), 5).cpu().numpy()
        for f in range(3):
            want = oracle.blur(fr[f].astype(np.float32), 5)
            assert np.array_equal(g[f].view(np.int32), want.view(np.int32)), (shape, f, np.abs(g[f] - want).max())
    fr = rng.integers(0, 4096, size=(40, 57), dtype=np.uint16)
    for k in (3, 7, 9):
        g = engine.blur_u16(torch.as_tensor(fr).cuda(), k).cpu().numpy()
        assert np.array_equal(g.view(np.int32), oracle.blur(fr.astype(np.float32), k).view(np.int32)), k


def test_blur_u16_quad_bitwise(gpu_lib, oracle):
    """The 5 x 5 blur of u16 frames with four pixels per lane (gauss5_quad_kernel: widths that are multiples of 4 --
    every camera format), bit for bit against the oracle:
    widths around the 256-column waves and the 1024-column workgroups (partly filled waves, halo loads of lanes 0 / 63,
    reflected columns at both image edges), heights around the 64-row pieces and below the kernel's halo, extreme
    values at the borders, several frames per call."""
    import torch
    from upsp_processing_amd import engine
    rng = np.random.default_rng(19)
    for shape in [(3, 8), (5, 12), (64, 256), (65, 260), (63, 252), (130, 1024), (37, 1028), (129, 1020), (70, 2052), (200, 516)]:
        fr = rng.integers(0, 4096, size=(3,) + shape, dtype=np.uint16)
        fr[0, 0, :] = 4095
        fr[0, :, 0] = 0
        fr[1, :, -1] = 4095
        fr[2, -1, :] = 0
        g = engine.blur_u16(torch.as_tensor(fr).cuda(), 5).cpu().numpy()
        for f in range(3):
            want = oracle.blur(fr[f].astype(np.float32), 5)
            assert np.array_equal(g[f].view(np.int32), want.view(np.int32)), (shape, f, np.abs(g[f] - want).max())


@pytest.mark.parametrize("interp", [1, 0])
def test_register_pixel(gpu_lib, oracle, interp):
    import torch
    from upsp_processing_amd import engine, synthetic as syn
    H, W = 192, 256
    fr = syn.synth_frames_numpy(5, H, W, seed=3, noise=2.0)
    ref = fr[0].astype(np.float32)
    d_ref = torch.as_tensor(ref).cuda()
    for f in range(1, 5):
        out_g, M_g, it_g = engine.register_pixel(d_ref, torch.as_tensor(fr[f].copy()).cuda(), interp=interp)
        out_o, M_o, it_o = oracle.register_pixel(ref, fr[f], interp=interp)
        assert it_g == it_o, (it_g, it_o)
        assert np.abs(M_g[:, :2] - M_o[:, :2]).max() <= 1e-4
        assert np.abs(M_g[:, 2] - M_o[:, 2]).max() <= 2e-3
        # the warp itself is exact integer arithmetic: same matrix -> same u16 frame
        assert np.array_equal(out_g.cpu().numpy(), oracle.warp_affine(fr[f], M_g, interp))


@pytest.mark.parametrize("H,W,shift", [(96, 131, (7.3, -4.6)), (48, 64, (1.4, 0.7)), (200, 300, (-15.2, 11.8)), (33, 47, (0.3, -0.2)),
                                       (1400, 1100, (3.3, -2.1))])
def test_register_pixel_band(gpu_lib, oracle, H, W, shift):
    """The ECC sums are taken by interior blocks (pixels farther than a band from every edge, no border handling) and
    band blocks (generic bilinear); the band follows the warp.  Odd image sizes, shifts of many pixels (wide bands, a band
    that swallows most of a small image), shear: same iteration count and warp as the oracle.  1400 x 1100: an image taller
    than 8 x 128 rows -- 5 column tiles x 11 row pieces of interior blocks instead of the 32 of a 1024^2 frame."""
    _register_band_case(oracle, H, W, shift)


def test_register_pixel_segments_of_32_rows(gpu_lib, oracle, monkeypatch):
    """UPSP_ECC_ONE_FLUSH=0: the interior blocks in the 32-row float segments of rounds 3-5 (what an image beyond 512
    interior blocks still takes) -- the same bars."""
    monkeypatch.setenv("UPSP_ECC_ONE_FLUSH", "0")
    _register_band_case(oracle, 200, 300, (-15.2, 11.8))
    _register_band_case(oracle, 1400, 1100, (3.3, -2.1))


def _register_band_case(oracle, H, W, shift):
    import torch
    from upsp_processing_amd import engine
    rng = np.random.default_rng(H * W)
    yy, xx = np.mgrid[0:H, 0:W].astype(np.float64)
    def scene(y, x):
        return (1800 + 700 * np.sin(2 * np.pi * 3 * x / W) * np.cos(2 * np.pi * 2 * y / H)
                + 400 * np.exp(-((x - 0.6 * W) ** 2 + (y - 0.4 * H) ** 2) / (0.02 * W * H)))
    ref16 = np.clip(scene(yy, xx) + rng.normal(0, 2, (H, W)), 0, 4095).astype(np.uint16)
    A = np.array([[1.0 + 2e-3, 1.5e-3, shift[0]], [-1e-3, 1.0 - 1e-3, shift[1]]])
    xs = A[0, 0] * xx + A[0, 1] * yy + A[0, 2]
    ys = A[1, 0] * xx + A[1, 1] * yy + A[1, 2]
    inp16 = np.clip(scene(ys, xs) + rng.normal(0, 2, (H, W)), 0, 4095).astype(np.uint16)
    ref = ref16.astype(np.float32)
    out_g, M_g, it_g = engine.register_pixel(torch.as_tensor(ref).cuda(), torch.as_tensor(inp16.copy()).cuda(), interp=1)
    out_o, M_o, it_o = oracle.register_pixel(ref, inp16, interp=1)
    assert it_g == it_o, (it_g, it_o)
    assert it_o >= 2
    assert np.abs(M_g[:, :2] - M_o[:, :2]).max() <= 1e-4
    assert np.abs(M_g[:, 2] - M_o[:, 2]).max() <= 2e-3
    assert np.array_equal(out_g.cpu().numpy(), oracle.warp_affine(inp16, M_g, 1))


def _patch_case():
    H, W = 400, 520
    yy, xx = np.mgrid[0:H, 0:W].astype(np.float64)
    f = lambda x, y: 1500 + 0.3 * x - 0.2 * y + 1e-3 * x * y + 2e-6 * x ** 2 * y - 1e-8 * y ** 3
    img = (f(xx, yy) + np.random.default_rng(0).normal(size=(H, W)) * 3).astype(np.float32)
    clusters = [disc_cluster(60, 50, 4, 7), disc_cluster(300, 200, 6, 9), disc_cluster(480, 360, 5, 8),
                dict(bx=[1, 2, 3], by=[1, 1, 1], ix=[2], iy=[2])]        # last: < 10 boundary points
    for cl in clusters[:3]:
        img[cl["iy"], cl["ix"]] *= 0.3                                    # fiducial discs
    return img, clusters


def _f64_fit(img, cl):
    x, y = np.asarray(cl["bx"], float), np.asarray(cl["by"], float)
    z = img[cl["by"], cl["bx"]].astype(np.float64)
    xm, ym = x.mean(), y.mean()
    mono = lambda x, y: np.stack([(y - ym) ** i * (x - xm) ** j for i in range(4) for j in range(4) if i + j <= 3], 1)
    coef = np.linalg.lstsq(mono(x, y), z, rcond=None)[0]
    return mono(np.asarray(cl["ix"], float), np.asarray(cl["iy"], float)) @ coef


def test_patch(gpu_lib, oracle):
    """PatchClusters::operator() (cpp/lib/patches.ipp:98-165) in the reference's arithmetic: float column-pivoted Householder QR
    on raw pixel coordinates (polyfit2D, :172-205) + polyval2D (:208-236).  The GPU runs the oracle's operations in the
    oracle's order (factorisation once on the host, Q^T z / back substitution / evaluation per frame and lane): BIT FOR BIT,
    the float-QR noise of the raw-coordinate fit included (reported below against a float64 fit)."""
    import torch
    from upsp_processing_amd import engine
    img, clusters = _patch_case()
    d = torch.as_tensor(img.copy()).cuda()
    engine.patch(d, clusters)
    out = d.cpu().numpy()
    out_o = oracle.patch_clusters(img, clusters)
    assert np.array_equal(out.view(np.int32), out_o.view(np.int32))
    changed = np.zeros(img.shape, bool)
    noise = 0.0
    for cl in clusters[:3]:
        noise = max(noise, float(np.abs(out[cl["iy"], cl["ix"]] - _f64_fit(img, cl)).max()) / 1500)
        changed[cl["iy"], cl["ix"]] = True
    assert np.array_equal(out[~changed], img[~changed])
    assert noise <= 3e-2          # the reference's own deviation from a well-conditioned fit (measured: see DESIGN.md section 2)
    print("patch: reference-arithmetic solve deviates from a float64 fit by %.2e relative" % noise)


def test_patch_many_frames_and_long_boundaries(gpu_lib, oracle):
    """The pipeline's shape: a wave per (cluster, 64 frames), lane = frame -- 70 frames (a full wave + a ragged one), a
    boundary longer than the LDS holds (scratch in global memory) and clusters whose boundary crosses another's interior
    (cluster order of the reference, patches.ipp:101): every frame bit for bit."""
    import torch
    from upsp_processing_amd import engine, _capi
    H, W, F = 300, 400, 70
    rng = np.random.default_rng(5)
    yy, xx = np.mgrid[0:H, 0:W].astype(np.float64)
    base = 900 + 0.5 * xx + 0.25 * yy + 1e-3 * xx * yy
    frames = np.stack([(base * (1 + 0.01 * f) + rng.normal(size=(H, W)) * 2).astype(np.float32) for f in range(F)])
    big = disc_cluster(200, 150, 30, 34)             # ring of ~800 boundary pixels (> 240 rows: global scratch)
    assert len(big["bx"]) > 240
    a, b = disc_cluster(60, 60, 4, 7), disc_cluster(70, 60, 4, 7)      # b's boundary crosses a's interior
    for clusters in ([disc_cluster(330, 80, 5, 8), big], [a, b]):
        d = torch.as_tensor(frames.copy()).cuda()
        engine.patch(d, clusters)
        out = d.cpu().numpy()
        for f in (0, 1, 63, 64, 69):
            want = oracle.patch_clusters(frames[f], clusters)
            assert np.array_equal(out[f].view(np.int32), want.view(np.int32)), f


def test_patch_pseudo_inverse_opt_in(gpu_lib, oracle, monkeypatch):
    """UPSP_PATCH_PINV=1: the centred, scaled double pseudo-inverse (a better-conditioned answer than the reference's, hence not
    the default): within 1e-5 of a float64 fit, and away from the reference-arithmetic result by that one's own noise."""
    import torch
    from upsp_processing_amd import engine
    monkeypatch.setenv("UPSP_PATCH_PINV", "1")
    img, clusters = _patch_case()
    d = torch.as_tensor(img.copy()).cuda()
    engine.patch(d, clusters)
    out = d.cpu().numpy()
    out_o = oracle.patch_clusters(img, clusters)
    dev = 0.0
    for cl in clusters[:3]:
        assert np.abs(out[cl["iy"], cl["ix"]] - _f64_fit(img, cl)).max() / 1500 <= 1e-5
        dev = max(dev, float(np.abs(out[cl["iy"], cl["ix"]] - out_o[cl["iy"], cl["ix"]]).max()) / 1500)
    assert dev <= 3e-2
    print("patch: pseudo-inverse solve deviates from the oracle by %.2e relative" % dev)


def oracle_loop(oracle, frames, ref, pix, clusters, first, registration, patch, filt, ksize):
    """psp_process.cpp:1771-1843 with the oracle pieces, one camera."""
    rows = []
    for i, fr in enumerate(frames):
        img, _ = oracle.fix_hot_pixels(fr)
        if registration and first + i > 0:
            img, M, it = oracle.register_pixel(ref, img)
            assert it > 0
        if patch:
            img = oracle.patch_clusters(img.astype(np.float32), clusters)
        if filt:
            img = oracle.blur(np.asarray(img, np.float32), ksize, box=(filt == 2))
        rows.append(oracle.project_frame(img, pix, None))
    return np.stack(rows)


@pytest.mark.parametrize("cfg", [dict(registration=0, patch=1, filter=1, filter_size=5),
                                 dict(registration=0, patch=0, filter=2, filter_size=3),
                                 dict(registration=1, patch=0, filter=0, filter_size=1),
                                 dict(registration=1, patch=1, filter=1, filter_size=3)])
def test_pipeline_with_stages(gpu_lib, oracle, cfg):
    import torch
    from upsp_processing_amd import engine, synthetic as syn
    H, W, F, n = 160, 224, 9, 4000
    frames = syn.synth_frames_numpy(F, H, W, seed=8, noise=2.0, hot=True)
    ref = frames[0].astype(np.float32)         # raw first frame, no hot-pixel fix (psp_process.cpp:2057)
    rng = np.random.default_rng(0)
    inner = rng.integers(12, H - 12, n) * W + rng.integers(12, W - 12, n)
    pix = inner.astype(np.int32)
    pix[::13] = -1
    clusters = [disc_cluster(50, 40, 4, 7), disc_cluster(150, 100, 5, 8)]
    pipe = engine.FramePipeline(1, W, H, n, **cfg)
    pipe.set_projection(0, pix)
    pipe.set_reference(0, ref)
    if cfg["patch"]:
        pipe.set_patches(0, clusters)
    warps = torch.zeros((F, 1, 6), dtype=torch.float32, device="cuda")
    rows_g = pipe.process(torch.as_tensor(frames.copy()).cuda(), 0, warps=warps if cfg["registration"] else None)
    rows_g = rows_g.cpu().numpy()
    rows_o = oracle_loop(oracle, frames, ref, pix, clusters, 0, cfg["registration"], cfg["patch"],
                         cfg["filter"], cfg["filter_size"])
    ok = pix >= 0
    assert np.isnan(rows_g[:, ~ok]).all()
    if not cfg["registration"]:
        # (the patch stage runs the reference's float QR operation for operation: bit for bit with it as well)
        assert np.array_equal(rows_g[:, ok].view(np.int32), rows_o[:, ok].view(np.int32))
    else:
        # registered frames: the warp matrices agree with the oracle's to 1e-4 / 2e-3 px (test_register_pixel*), and the
        # warp itself is exact integer arithmetic -- so the oracle chain fed with the GPU's matrices must give the
        # GPU's rows bit for bit, patch stage or not
        w = warps.cpu().numpy()[:, 0]
        assert np.array_equal(w[0], [1, 0, 0, 0, 1, 0])     # frame 0 is never registered
        rows_m = []
        for i, fr in enumerate(frames):
            img, _ = oracle.fix_hot_pixels(fr)
            if i > 0:
                _, M_o, _ = oracle.register_pixel(ref, img)
                assert np.abs(w[i].reshape(2, 3)[:, :2] - M_o[:, :2]).max() <= 1e-4
                assert np.abs(w[i].reshape(2, 3)[:, 2] - M_o[:, 2]).max() <= 2e-3
                img = oracle.warp_affine(img, w[i].reshape(2, 3), 1)
            if cfg["patch"]:
                img = oracle.patch_clusters(img.astype(np.float32), clusters)
            if cfg["filter"]:
                img = oracle.blur(np.asarray(img, np.float32), cfg["filter_size"], box=(cfg["filter"] == 2))
            rows_m.append(oracle.project_frame(img, pix, None))
        rows_m = np.stack(rows_m)
        assert np.array_equal(rows_g[:, ok].view(np.int32), rows_m[:, ok].view(np.int32))
        # and the oracle's own chain (its matrices): within one 1/32-px step of the warp coordinates
        d = np.abs(rows_g[:, ok] - rows_o[:, ok])
        fixed = np.stack([oracle.fix_hot_pixels(fr)[0] for fr in frames]).astype(np.int32)
        step = max(np.abs(np.diff(fixed, axis=1)).max(), np.abs(np.diff(fixed, axis=2)).max())
        # (with a patch stage a pixel one warp step apart moves the float-QR fit of its cluster: its own noise on top)
        assert d.max() <= 2.0 * step / 32.0 + 1.0 + (1e-2 * 1800 if cfg["patch"] else 0) and d.mean() <= 0.5

def test_registration_full_size_1024(gpu_lib, oracle):
    """configs[2] at its own image size: 1024 x 1024 frames through FramePipeline(registration=1) against
    oracle.register_pixel + project_frame (cpp/lib/registration.cpp:32-81, cpp/exec/psp_process.cpp:1776-1795).

    9 frames of the bench's image model (sub-pixel jitter, hot pixels) + one frame moved by ~9 px with shear
    (wide border band of the ECC sums: the interior / band split of ecc_sums2_kernel takes its ranges from the
    image size and the warp) + one moved by 2.4 px.  Bars:
      ECC warp matrix        max|dM| <= 1e-4 (linear part), <= 2e-3 px (translation), per frame
      ECC iteration count    identical, per frame
      warped u16 frame       bit-exact for the GPU's own matrix (exact integer arithmetic)
      series rows            bit-exact vs project_frame(warpAffine(frame, M_gpu)); vs the oracle's own chain
                             (its M) within ONE STEP of the warp's 1/32-px fixed-point coordinates in x and y:
                             |dI| <= 2 x (largest step between neighbouring pixels of the frame) / 32 + 1 count
                             of rounding, mean <= 0.25"""
    import torch
    from upsp_processing_amd import engine, synthetic as syn
    H = W = 1024
    F, n = 11, 60000
    frames = syn.synth_frames_numpy(F - 2, H, W, seed=11, hot=True)
    ref = frames[0].astype(np.float32)
    # two frames with real model motion: frame 0's scene resampled through a known affine map (border pixels
    # come in as 0 -> the mask and the band paths are exercised), fresh noise on top
    rng = np.random.default_rng(12)
    moved = []
    for A in (np.array([[1.0 + 2.5e-3, 2e-3, 8.7], [-1.5e-3, 1.0 - 2e-3, -9.4]], np.float32),
              np.array([[1.0, -4e-4, -2.4], [6e-4, 1.0, 1.3]], np.float32)):
        m = oracle.warp_affine(frames[0], A, 1).astype(np.float64) + rng.normal(0, 4, (H, W))
        moved.append(np.clip(np.rint(m), 0, 4095).astype(np.uint16))
    frames = np.concatenate([frames[:5], moved[0][None], frames[5:], moved[1][None]])
    assert frames.shape[0] == F
    inner = rng.integers(16, H - 16, n) * W + rng.integers(16, W - 16, n)
    pix = inner.astype(np.int32)
    pix[::17] = -1
    pipe = engine.FramePipeline(1, W, H, n, registration=1)
    pipe.set_projection(0, pix)
    pipe.set_reference(0, ref)
    warps = torch.zeros((F, 1, 6), dtype=torch.float32, device="cuda")
    iters = torch.full((F, 1), -1, dtype=torch.int32, device="cuda")
    d = torch.as_tensor(frames.copy()).cuda()
    rows_g = pipe.process(d, 0, warps=warps, ecc_iters=iters).cpu().numpy()
    w = warps.cpu().numpy()[:, 0].reshape(F, 2, 3)
    it_g = iters.cpu().numpy()[:, 0]
    ok = pix >= 0
    assert np.isnan(rows_g[:, ~ok]).all()
    assert np.array_equal(w[0].reshape(-1), [1, 0, 0, 0, 1, 0]) and it_g[0] == 0      # frame 0 is never registered
    st = pipe.ecc_stats()
    assert st["frames"] == F and st["frame_iterations"] == int(it_g.sum())
    worst = [0.0, 0.0, 0.0]
    for f in range(F):
        img, _ = oracle.fix_hot_pixels(frames[f])
        assert np.array_equal(d[f].cpu().numpy(), img), f                              # repaired in place, like the reference
        if f == 0:
            assert np.array_equal(rows_g[0, ok].view(np.int32), oracle.project_frame(img, pix, None)[ok].view(np.int32))
            continue
        out_o, M_o, it_o = oracle.register_pixel(ref, img)
        assert it_g[f] == it_o, (f, it_g[f], it_o)
        dl, dt = np.abs(w[f][:, :2] - M_o[:, :2]).max(), np.abs(w[f][:, 2] - M_o[:, 2]).max()
        assert dl <= 1e-4 and dt <= 2e-3, (f, dl, dt)
        # same matrix -> same u16 frame -> same rows, bit for bit
        warped = oracle.warp_affine(img, w[f], 1)
        want = oracle.project_frame(warped, pix, None)
        assert np.array_equal(rows_g[f, ok].view(np.int32), want[ok].view(np.int32)), f
        # the whole oracle chain with its own matrix
        dd = np.abs(rows_g[f, ok] - oracle.project_frame(out_o, pix, None)[ok])
        fx = img.astype(np.int32)
        step = max(np.abs(np.diff(fx, axis=0)).max(), np.abs(np.diff(fx, axis=1)).max())
        assert dd.max() <= 2.0 * step / 32.0 + 1.0 and dd.mean() <= 0.25, (f, dd.max(), step, dd.mean())
        worst = [max(worst[0], dl), max(worst[1], dt), max(worst[2], float(dd.max()))]
    assert it_g[5] >= 3                                                                 # the 9-px frame really iterates
    # node-major series wanted: the warp writes the active pixels straight into the compact buffer and pass B writes whole
    # rows (the bench's schedule) -- same bits as the frame-major rows above, same accumulators as the gather's
    acc_a = [a.clone() for a in pipe.accumulators()]
    for split in (F, 7):
        pipe2 = engine.FramePipeline(1, W, H, n, registration=1)
        pipe2.set_projection(0, pix)
        pipe2.set_reference(0, ref)
        rt = torch.full((n, engine.series_ld(F)), -7.0, dtype=torch.float32, device="cuda")
        d2 = torch.as_tensor(frames.copy()).cuda()
        w2 = torch.zeros((F, 1, 6), dtype=torch.float32, device="cuda")
        for f0 in range(0, F, split):
            f1 = min(F, f0 + split)
            pipe2.process(d2[f0:f1], f0, rows_t=rt[:, :F], col0=f0, want_rows=False, warps=w2[f0:f1])
        got = rt[:, :F].cpu().numpy().T
        assert np.isnan(got[:, ~ok]).all() and (rt[:, F:] == -7.0).all() and torch.equal(d2, d)
        if split == F:
            # one call: the same sub-batch as above, so the same warps -> the same bits, the same accumulators
            assert torch.equal(w2, warps)
            assert np.array_equal(got[:, ok].view(np.int32), rows_g[:, ok].view(np.int32))
            for a, b in zip(pipe2.accumulators(), acc_a):
                aa, bb = a.cpu().numpy(), b.cpu().numpy()
                assert np.array_equal(np.isnan(aa), np.isnan(bb)) and np.array_equal(aa[~np.isnan(aa)], bb[~np.isnan(bb)])
        else:
            # calls of 7 frames into column blocks of the same rows: other sub-batches, the SAME warps bit for bit -- the
            # blocks (and with them the float segments of the column sums) are cut by the image geometry alone, not by how
            # many frames of the sub-batch are still iterating
            ww = w2.cpu().numpy()[:, 0].reshape(F, 2, 3)
            assert np.array_equal(ww.view(np.int32), w.view(np.int32))
            for f in range(1, F):
                img, _ = oracle.fix_hot_pixels(frames[f])
                want = oracle.project_frame(oracle.warp_affine(img, ww[f], 1), pix, None)
                assert np.array_equal(got[f, ok].view(np.int32), want[ok].view(np.int32)), (split, f)
            s2 = pipe2.accumulators()[0].cpu().numpy()
            assert np.array_equal(s2[ok], got[:, ok].astype(np.float64).sum(0))
    print("1024^2 registration: iterations %s, worst |dM| %.2e, |dt| %.2e px, |dI| %.2f" % (it_g.tolist(), *worst))


@pytest.mark.parametrize("reg_batch", ["64", "128", None])
def test_registration_sub_batches_look_ahead(gpu_lib, oracle, monkeypatch, reg_batch):
    """(UPSP_REG_BATCH: frames per sub-batch of the streamed registration path -- 512 by default; 64 and 128 put several
    sub-batches into the calls below.  The sums of a frame do not depend on its neighbours: the same bits whatever the size.)
    More frames than one sub-batch through the streamed registration path: the hot-pixel repair and the pre-blur of
    sub-batch k + 1 are enqueued while the host waits for sub-batch k's "frames still iterating" (two blurred-frame buffers).
    Same bits as calls of one sub-batch each (no look-ahead: nothing to look ahead to) -- series, warps, iteration counts,
    accumulators, repaired frames -- over several calls (buffers re-used), with hot pixels in frames of every sub-batch; and
    3 frames against the oracle."""
    import torch
    from upsp_processing_amd import engine, synthetic as syn
    if reg_batch:
        monkeypatch.setenv("UPSP_REG_BATCH", reg_batch)
    H, W, F, n = 96, 160, 200, 3000
    frames = syn.synth_frames_numpy(F, H, W, seed=21, hot=True)
    rng = np.random.default_rng(22)
    for f in (3, 64, 65, 130, 199):                      # hot pixels in every sub-batch of 64
        frames[f, rng.integers(2, H - 2), rng.integers(2, W - 2)] = 4090
    ref = frames[0].astype(np.float32)
    pix = (rng.integers(8, H - 8, n) * W + rng.integers(8, W - 8, n)).astype(np.int32)
    pix[::13] = -1
    ok = pix >= 0
    out = {}
    for mode, calls in (("look-ahead", ((0, 150), (150, 200))), ("one sub-batch per call", ((0, 64), (64, 100), (100, 150), (150, 200)))):
        pipe = engine.FramePipeline(1, W, H, n, registration=1)
        pipe.set_projection(0, pix)
        pipe.set_reference(0, ref)
        ld = engine.series_ld(F)
        rt = torch.full((n, ld), -3.0, dtype=torch.float32, device="cuda")
        d = torch.as_tensor(frames.copy()).cuda()
        w = torch.zeros((F, 1, 6), dtype=torch.float32, device="cuda")
        it = torch.full((F, 1), -1, dtype=torch.int32, device="cuda")
        for f0, f1 in calls:
            pipe.process(d[f0:f1], f0, rows_t=rt[:, :F], col0=f0, want_rows=False, warps=w[f0:f1], ecc_iters=it[f0:f1])
        torch.cuda.synchronize()
        out[mode] = (rt.cpu().numpy(), w.cpu().numpy(), it.cpu().numpy(), [a.cpu().numpy() for a in pipe.accumulators()], d.cpu().numpy())
        pipe.close()
    a, b = out["look-ahead"], out["one sub-batch per call"]
    assert np.array_equal(a[0].view(np.int32), b[0].view(np.int32)) and np.array_equal(a[1].view(np.int32), b[1].view(np.int32))
    assert np.array_equal(a[2], b[2]) and np.array_equal(a[4], b[4])
    for x, y in zip(a[3], b[3]):
        assert np.array_equal(np.isnan(x), np.isnan(y)) and np.array_equal(x[~np.isnan(x)], y[~np.isnan(y)])
    for f in (64, 65, 199):
        img, _ = oracle.fix_hot_pixels(frames[f])
        assert np.array_equal(a[4][f], img)
        _, M_o, it_o = oracle.register_pixel(ref, img)
        M_g = a[1][f, 0].reshape(2, 3)
        assert a[2][f, 0] == it_o and np.abs(M_g[:, :2] - M_o[:, :2]).max() <= 1e-4 and np.abs(M_g[:, 2] - M_o[:, 2]).max() <= 2e-3
        want = oracle.project_frame(oracle.warp_affine(img, M_g, 1), pix, None)
        assert np.array_equal(a[0][ok, f].view(np.int32), want[ok].view(np.int32))


@pytest.mark.parametrize("H,W,F", [(96, 160, 70), (300, 250, 24), (131, 1030, 22)])
def test_ecc_fused_blur_same_frames_and_bars(gpu_lib, oracle, monkeypatch, H, W, F):
    """The streamed registration loop blurs the frames and takes the sums of the ECC's identity iteration in ONE pass
    (ecc_blur_ident_kernel: 58-column strips per wave, reflected loads at the image edges, hot-pixel scan on the way, a second
    pass over the frames the repair changed).  Against the two-kernel path (UPSP_ECC_FUSED_BLUR=0):
      * UPSP_ECC_FUSED_BLUR=2 keeps the pass's blurred frames and drops its sums: series, warps, iteration counts and repaired
        frames must be the SAME BITS -- i.e. the blurred frames are bit-identical, image edges and repaired neighbourhoods included;
      * the default (sums from the pass): same iteration counts, warps within the oracle's bars of the two-kernel path (the sums
        are taken in another order), and the oracle's bars themselves on frames with and without hot pixels.
    Widths that leave a partial strip, heights that leave a partial row piece, more than one row piece and sub-batch."""
    import torch
    from upsp_processing_amd import engine, synthetic as syn
    monkeypatch.setenv("UPSP_REG_BATCH", "32")
    n = 2500
    frames = syn.synth_frames_numpy(F, H, W, seed=H + W, hot=True)
    rng = np.random.default_rng(H)
    hot_frames = (1, F // 2, F - 1)
    for f in hot_frames:                                   # one in the interior, one at an image edge, one in a corner
        frames[f, rng.integers(2, H - 2), rng.integers(2, W - 2)] = 4090
    frames[hot_frames[1], 0, rng.integers(2, W - 2)] = 4095
    frames[hot_frames[2], H - 1, W - 1] = 4095
    # at the seams of the wave items (58-column strips, 128-row pieces): the second pass re-runs the workgroups on BOTH sides
    for f, (y, x) in zip(range(3, 12), ((min(127, H - 1), 57), (min(128, H - 1), 58), (min(130, H - 1), 55), (5, 60), (min(125, H - 1), 115),
                                        (min(126, H - 1), 118), (3, 0), (0, 59), (min(129, H - 1), W - 1))):
        frames[f, y, min(x, W - 1)] = 4093
    frames[12:20, H // 3, W // 3] = 4091                   # a stuck pixel: the same one in consecutive frames
    ref = frames[0].astype(np.float32)
    pix = (rng.integers(0, H, n) * W + rng.integers(0, W, n)).astype(np.int32)
    out = {}
    for mode in ("0", "2", "1"):
        monkeypatch.setenv("UPSP_ECC_FUSED_BLUR", mode)
        pipe = engine.FramePipeline(1, W, H, n, registration=1)
        pipe.set_projection(0, pix)
        pipe.set_reference(0, ref)
        rt = torch.full((n, engine.series_ld(F)), -3.0, dtype=torch.float32, device="cuda")
        d = torch.as_tensor(frames.copy()).cuda()
        w = torch.zeros((F, 1, 6), dtype=torch.float32, device="cuda")
        it = torch.full((F, 1), -1, dtype=torch.int32, device="cuda")
        pipe.process(d, 0, rows_t=rt[:, :F], want_rows=False, warps=w, ecc_iters=it)
        torch.cuda.synchronize()
        out[mode] = (rt.cpu().numpy()[:, :F], w.cpu().numpy()[:, 0].reshape(F, 2, 3), it.cpu().numpy()[:, 0], d.cpu().numpy())
        pipe.close()
    a, b, c = out["0"], out["2"], out["1"]
    assert np.array_equal(a[3], b[3]) and np.array_equal(a[3], c[3])                    # repaired frames
    assert np.array_equal(a[0].view(np.int32), b[0].view(np.int32)) and np.array_equal(a[1].view(np.int32), b[1].view(np.int32))
    assert np.array_equal(a[2], b[2])
    assert np.array_equal(a[2], c[2]), (a[2], c[2])
    # (the float rounding of a sum moves a warp by 1e-7 .. 1e-4 px through the reference's float 6 x 6 solve, whichever kernel took
    #  the sums -- tools/r06_fused_dbg.py prints both paths against the oracle; frames that oscillate for tens of iterations amplify it)
    few = a[2] <= 6
    assert few.sum() >= F // 2
    assert np.abs(a[1][few][:, :, :2] - c[1][few][:, :, :2]).max() <= 1e-4 and np.abs(a[1][few][:, :, 2] - c[1][few][:, :, 2]).max() <= 2e-3
    for f in (hot_frames[0], hot_frames[2], 2, 3, 4, 8, 11, 13):
        img, _ = oracle.fix_hot_pixels(frames[f])
        assert np.array_equal(c[3][f], img)
        _, M_o, it_o = oracle.register_pixel(ref, img)
        assert c[2][f] == it_o and np.abs(c[1][f][:, :2] - M_o[:, :2]).max() <= 1e-4 and np.abs(c[1][f][:, 2] - M_o[:, 2]).max() <= 2e-3
        want = oracle.project_frame(oracle.warp_affine(img, c[1][f], 1), pix, None)
        assert np.array_equal(c[0][:, f].view(np.int32), want.view(np.int32))


def test_ecc_lds_taps_same_bits(gpu_lib, monkeypatch):
    """The general ECC iteration takes its 12 source taps per pixel from an LDS-staged tile of the source frame (one float
    segment of 32 rows x 256 columns at a time) instead of 8 load instructions per pixel: the same floats through the same
    arithmetic in the same order, so warps, iteration counts and series are bit-identical to the direct loads
    (UPSP_ECC_DIRECT=1, the path of segments whose footprint does not fit the tile) -- on frames with sub-pixel jitter, with
    rotation and scale large enough that some segments fall back by themselves, with a width that is not a multiple of four
    (every segment direct), widths that leave a partial column tile, shifts of several pixels (wide band), a second call."""
    import torch
    from upsp_processing_amd import engine, synthetic as syn
    rng = np.random.default_rng(41)
    from scipy.ndimage import map_coordinates
    for (H, W, F, lin, shift) in ((203, 300, 16, 0.0, 2.5), (256, 512, 12, 4e-3, 2.5), (130, 1028, 10, 1.5e-2, 2.5),
                                  (300, 260, 10, 4e-2, 1.0), (97, 520, 8, 1e-3, 9.0), (150, 301, 8, 2e-3, 2.5)):
        base = syn.synth_frames_numpy(1, H, W, seed=H, noise=0.0)[0].astype(np.float64)
        yy, xx = np.mgrid[0:H, 0:W].astype(np.float64)
        frames = np.empty((F, H, W), np.uint16)
        frames[0] = base
        for f in range(1, F):
            a = rng.uniform(-lin, lin, 4)
            sh = rng.uniform(-shift, shift, 2)
            ys = a[2] * xx + (1 + a[3]) * yy + sh[1]
            xs = (1 + a[0]) * xx + a[1] * yy + sh[0]
            frames[f] = np.clip(np.rint(map_coordinates(base, [ys, xs], order=1, mode="nearest") + rng.normal(0, 3, (H, W))), 0, 4095)
        n = 2000
        pix = (rng.integers(8, H - 8, n) * W + rng.integers(8, W - 8, n)).astype(np.int32)
        out = {}
        for mode in ("1", "0"):
            monkeypatch.setenv("UPSP_ECC_DIRECT", mode)
            pipe = engine.FramePipeline(1, W, H, n, registration=1)
            pipe.set_projection(0, pix)
            pipe.set_reference(0, frames[0].astype(np.float32))
            w = torch.zeros((F, 1, 6), dtype=torch.float32, device="cuda")
            it = torch.full((F, 1), -1, dtype=torch.int32, device="cuda")
            rows = pipe.process(torch.as_tensor(frames.copy()).cuda(), 0, warps=w, ecc_iters=it)
            rows2 = pipe.process(torch.as_tensor(frames[: F // 2].copy()).cuda(), 0)
            out[mode] = (w.cpu().numpy(), it.cpu().numpy(), rows.cpu().numpy(), rows2.cpu().numpy())
            pipe.close()
        b = out["1"]
        assert int(b[1].max()) >= 2                      # general iterations did run
        for x, y in zip(out["0"], b):
            assert np.array_equal(x.view(np.int32), y.view(np.int32)), (H, W)
