"""Host-side pieces of bench.py that need no GPU: the per-pixel rays of the `pixel_rays` leg, the CPU share."""
import numpy as np


def test_pixel_rays_project_back_to_their_pixel(oracle):
    """bench.pixel_rays: the ray through pixel (u, v) -- a point on it projects back to (u, v) with the oracle's
    cv::projectPoints restatement (the bench camera has no distortion), for an oblique camera too."""
    import bench
    from upsp_processing_amd import synthetic as syn
    for kw in (dict(center=(0, 0, 20), half_extent=6.0), dict(center=(3, -2, 15), half_extent=4.0, azimuth_deg=40.0)):
        size = 64
        cd = syn.pinhole_camera(size, size, **kw)
        org, dirs = bench.pixel_rays(cd, size)
        assert dirs.shape == (size * size, 3) and np.allclose(np.linalg.norm(dirs, axis=1), 1.0, atol=1e-6)
        cam = oracle.make_camera(cd["K"], cd["dist"], cd["R"], cd["t"], size, size)
        assert np.allclose(org, oracle.cam_center(cam), atol=1e-5)
        idx = np.arange(0, size * size, 97)
        pts = org[None, :] + dirs[idx] * np.float32(11.0)
        uv = oracle.project_points(cam, pts)
        v, u = np.divmod(idx, size)
        assert np.abs(uv[:, 0] - u).max() < 2e-3 and np.abs(uv[:, 1] - v).max() < 2e-3


def test_usable_cpus_is_positive():
    import bench
    n = bench.usable_cpus()
    assert isinstance(n, int) and n >= 1


def test_rays_entering_counts_the_lines_through_the_box():
    """bench.rays_entering (the `rays_entered` figure of the pixel-ray legs): slab test of the ray's LINE, axis-parallel
    rays included."""
    import bench
    org = np.array([0.0, 0.0, 10.0])
    d = np.array([[0, 0, -1.0], [0, 0, 1.0], [1, 0, 0.0], [0.05, 0, -1.0], [0.5, 0, -1.0], [0, 0.0999, -1.0]])
    # box [-1, 1]^3: straight down enters; straight up is the same LINE (the reference's box test is a line test);
    # parallel to x at z = 10 misses; slightly tilted enters (x = 0.45..0.55 inside the slab), strongly tilted misses
    assert bench.rays_entering(org, d, [-1, -1, -1], [1, 1, 1]) == 4
