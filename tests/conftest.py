import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as orc
    orc.build()
    return orc


@pytest.fixture(scope="session")
def gpu_lib():
    """The product library; GPU tests fail (not skip) when it is missing."""
    import torch
    from upsp_processing_amd import _capi
    assert torch.cuda.is_available(), "GPU test selected but no HIP device is visible"
    L = _capi.lib()
    info = _capi.device_info()
    assert info["arch"].startswith("gfx950"), info
    return L


@pytest.fixture(scope="session")
def fml(oracle):
    """fml_tc3_volume.grid fixture of the reference's test suite, prepared like
    test/python/test_visibility.py setUpClass."""
    import numpy as np
    import refdata
    verts, inds = refdata.fml_grid()
    prims = refdata.package_primitives(verts, inds).astype(np.float32)
    nodes, norms, faces, fn = refdata.tvecs_and_norms(verts, inds)
    rmat, tvec, cm, dist = refdata.read_camera_tunnel_cal(
        os.path.join(refdata.GOLDEN, "camera01_35_6.json"), (512, 1024))
    cam_t = -(rmat.T @ tvec)
    return dict(prims=prims, nodes=nodes, norms=norms, nfaces=faces.shape[0], rmat=rmat,
                tvec=tvec, cm=cm, dist=dist, cam_t=cam_t)
