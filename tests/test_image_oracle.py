"""CPU: sanity of the OpenCV / Eigen restatements in oracle/image_oracle.c.
PARITY UNPINNED -- the reference holds no test and no golden data for register_pixel,
the filters or PatchClusters; these tests pin the restatements on independent
implementations (scipy.ndimage, numpy lstsq) and on self-consistency properties."""
import numpy as np
import pytest
from scipy import ndimage


def disc_cluster(cx, cy, r_in, r_out):
    xs, ys = np.meshgrid(np.arange(cx - r_out - 2, cx + r_out + 3), np.arange(cy - r_out - 2, cy + r_out + 3))
    r = np.hypot(xs - cx, ys - cy)
    b, i = (r > r_in) & (r <= r_out), r <= r_in
    return dict(bx=xs[b], by=ys[b], ix=xs[i], iy=ys[i])


def test_gaussian_and_box_blur(oracle):
    img = np.random.default_rng(0).normal(size=(37, 53)).astype(np.float32)
    for k, kern in ((3, [.25, .5, .25]), (5, [.0625, .25, .375, .25, .0625]),
                    (7, [.03125, .109375, .21875, .28125, .21875, .109375, .03125])):
        ref = ndimage.correlate1d(ndimage.correlate1d(img.astype(np.float64), kern, axis=1, mode="mirror"),
                                  kern, axis=0, mode="mirror")
        assert np.abs(oracle.blur(img, k) - ref).max() < 1e-6
    g9 = oracle.blur(img, 9)
    sigma = 0.3 * ((9 - 1) * 0.5 - 1) + 0.8
    x = np.arange(9) - 4
    kern = np.exp(-0.5 * x * x / sigma ** 2); kern /= kern.sum()
    ref = ndimage.correlate1d(ndimage.correlate1d(img.astype(np.float64), kern, axis=1, mode="mirror"), kern, axis=0, mode="mirror")
    assert np.abs(g9 - ref).max() < 1e-6
    for k in (3, 5):
        assert np.abs(oracle.blur(img, k, box=True) - ndimage.uniform_filter(img.astype(np.float64), k, mode="mirror")).max() < 1e-6
    const = np.full((9, 11), 3.25, np.float32)
    assert np.array_equal(oracle.blur(const, 5), const)       # preMask stays exactly 1 in ECC


def test_warp_affine(oracle):
    rng = np.random.default_rng(1)
    img = rng.integers(0, 4096, size=(40, 60)).astype(np.uint16)
    ident = np.array([[1, 0, 0], [0, 1, 0]], np.float32)
    assert np.array_equal(oracle.warp_affine(img, ident, 1), img)
    assert np.array_equal(oracle.warp_affine(img, ident, 0), img)
    shift = np.array([[1, 0, 3], [0, 1, -2]], np.float32)      # dst(x,y) = src(x+3, y-2)
    w = oracle.warp_affine(img, shift, 1)
    assert np.array_equal(w[2:, :-3], img[:-2, 3:]) and (w[:2] == 0).all() and (w[:, -3:] == 0).all()
    half = np.array([[1, 0, 0.5], [0, 1, 0]], np.float32)
    w = oracle.warp_affine(img.astype(np.float32), half, 1)
    assert np.allclose(w[:, :-1], 0.5 * (img[:, :-1].astype(np.float32) + img[:, 1:]))
    # fractional offsets are quantised to 1/32 px
    q = np.array([[1, 0, 0.26], [0, 1, 0]], np.float32)
    w = oracle.warp_affine(img.astype(np.float32), q, 1)
    assert np.allclose(w[:, :-1], (1 - 8 / 32) * img[:, :-1] + (8 / 32) * img[:, 1:], rtol=1e-6)


def test_ecc_recovers_known_affine(oracle):
    from upsp_processing_amd import synthetic as syn
    H, W = 192, 256
    fr = syn.synth_frames_numpy(4, H, W, seed=3, noise=2.0)
    A = syn.frame_params(4, 3)
    to3 = lambda a: np.vstack([a, [0, 0, 1]])
    ref = fr[0].astype(np.float32)
    for f in (1, 2, 3):
        out, M, it = oracle.register_pixel(ref, fr[f])
        expect = (np.linalg.inv(to3(A[f])) @ to3(A[0]))[:2]     # Input(W x) = Template(x)
        assert 1 <= it <= 50
        assert np.abs(M[:, :2] - expect[:, :2]).max() < 5e-4
        assert np.abs(M[:, 2] - expect[:, 2]).max() < 0.08
        # registered frame is closer to the template than the raw frame
        inner = (slice(8, -8), slice(8, -8))
        e0 = np.abs(fr[f][inner].astype(np.float32) - ref[inner]).mean()
        e1 = np.abs(out[inner].astype(np.float32) - ref[inner]).mean()
        assert e1 < e0
    # identical images: converges at once with the identity
    _, M, it = oracle.register_pixel(ref, fr[0])
    assert np.abs(M - np.array([[1, 0, 0], [0, 1, 0]])).max() < 1e-5


def test_polyfit_and_patch(oracle):
    cl = disc_cluster(110, 210, 6, 9)
    x, y = np.asarray(cl["bx"]), np.asarray(cl["by"])
    f = lambda x, y: 1500 + 0.3 * x - 0.2 * y + 1e-3 * x * y + 2e-6 * x ** 2 * y
    poly, rank = oracle.polyfit2d(x, y, f(x, y))
    assert 1 <= rank <= 10
    # the reference's float QR on raw pixel coordinates is ill-conditioned:
    # it reproduces the surface to ~5e-3 relative, not better
    img = np.zeros((300, 300), np.float32)
    yy, xx = np.mgrid[0:300, 0:300]
    img[:] = f(xx, yy)
    img[cl["iy"], cl["ix"]] = 0
    out = oracle.patch_clusters(img, [cl, dict(bx=[1, 2], by=[1, 2], ix=[5], iy=[5])])
    rel = np.abs(out[cl["iy"], cl["ix"]] - f(cl["ix"], cl["iy"])) / 1500
    assert rel.max() < 1e-2
    assert out[5, 5] == img[5, 5]                         # < 10 boundary points: cluster skipped
    mask = np.ones_like(img, bool); mask[cl["iy"], cl["ix"]] = False
    assert np.array_equal(out[mask], img[mask])           # only interior pixels change
