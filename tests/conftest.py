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


def build_rccl_shim():
    """tests/shim/librccl_shim.so: the stand-in RCCL that lets several rank PROCESSES share one GPU (test infrastructure;
    libupsp_gpu.so binds it only when UPSP_RCCL_LIBRARY names it)."""
    import subprocess
    src = os.path.join(ROOT, "tests", "shim", "rccl_shim.cpp")
    out = os.path.join(ROOT, "tests", "shim", "librccl_shim.so")
    if not os.path.exists(out) or os.path.getmtime(out) < os.path.getmtime(src):
        subprocess.check_call(["/opt/rocm/bin/hipcc", "-O1", "-std=c++17", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out, src, "-lrt"])
    return out


@pytest.fixture(scope="session")
def rccl_shim():
    return build_rccl_shim()


def one_gpu_ranks_env(shim):
    """Environment of a multi-rank child job on a ONE-GPU box: every rank on cuda:0, torch.distributed's rendezvous over gloo,
    the library's exchange through the stand-in RCCL.  Two or more GPUs: nothing (real RCCL, one rank per GPU)."""
    import torch
    if torch.cuda.device_count() >= 2:
        return {}
    return {"UPSP_BACKEND": "gloo", "UPSP_ONE_GPU": "1", "UPSP_BENCH_BACKEND": "gloo", "UPSP_BENCH_ONE_GPU": "1", "UPSP_RCCL_LIBRARY": shim}


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
