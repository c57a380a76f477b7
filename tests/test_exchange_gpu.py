"""GPU: the rank-to-rank exchanges behind the C ABI (include/upsp_gpu.h section 3b), driven by a C++ program the way
a C++ psp_process would (cpp/exec/psp_process.cpp:707-771, 1866-1872).

* `local W`: W ranks in one process on one GPU -- device-to-device copies stand in for the links, everything else
  (apportion, chunk boundaries, ragged blocks, packed rows, the f32 / u16 / 12-bit wire formats, NaN rows, byte
  counts) is the code a multi-GPU run executes; W = 1, 2, 3, 5 with node and frame counts that divide by none of them;
* `rccl1`: the same through RCCL itself in a one-rank communicator (grouped ncclSend / ncclRecv to self, ncclAllReduce);
* `ranks`: W rank PROCESSES on this GPU, the RCCL branch of csrc/exchange.hip (grouped ncclSend / ncclRecv per peer,
  ncclAllReduce) bound to the tests' stand-in RCCL through UPSP_RCCL_LIBRARY (tests/shim/rccl_shim.cpp): every wire of the
  node rows, the pixel-series mode placed and in place, two exchanges in turn on one communicator; W = 2, 3, 5 (the pool
  allows six processes on a GPU, this one included);
* two real RCCL ranks where two GPUs are visible (skipped on a one-GPU box: RCCL refuses two ranks on one device)."""
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def exe(tmp_path_factory, gpu_lib):
    out = str(tmp_path_factory.mktemp("xchg") / "exchange_test")
    libdir = os.path.join(ROOT, "upsp_processing_amd", "lib")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-O1", "-std=c++17", "-I" + os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "cpp", "exchange_test.cpp"), "-o", out,
                           "-L" + libdir, "-lupsp_gpu", "-Wl,-rpath," + libdir])
    return out


@pytest.mark.parametrize("world", [1, 2, 3, 5])
def test_exchange_local_ranks(exe, world):
    r = subprocess.run([exe, "local", str(world)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    lines = r.stdout.strip().splitlines()
    assert len(lines) == 3 and all(" ok," in l for l in lines), r.stdout
    sent = [int(l.split(", ")[1].split()[0]) for l in lines]
    if world == 1:
        assert sent == [0, 0, 0]
    else:
        assert sent[0] == 2 * sent[1] and sent[2] < sent[1]        # f32 : u16 : 12 bit


@pytest.mark.parametrize("world", [1, 2, 3, 5])
def test_exchange_pixel_series_local_ranks(exe, world):
    """The pixel-series mode (upsp_exchange_set_pixels / submit_pixels / finish_pixels): every destination receives the
    active pixels its node slice reads, each once, and runs pass B itself -- series, NaN rows and COMPLETE accumulators
    (after the all-reduce of slices that are zero elsewhere) against the closed form, u16 and 12-bit wire; fewer pixel
    rows than travelling nodes cross the links."""
    r = subprocess.run([exe, "pixels", str(world)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    lines = r.stdout.strip().splitlines()
    # (two shapes x two wires: 517 frames in 3 ragged chunks -- placed --, and 240 frames as one block per peer, which the
    #  owner's pass B reads where it arrived)
    assert len(lines) == 4 and all(" ok," in l for l in lines), r.stdout
    rows = int(lines[0].split(", ")[2].split()[0])
    assert rows <= 211 * world and rows < 803            # <= A pixels per destination, fewer than the travelling nodes


@pytest.mark.parametrize("self_rccl", ["0", "1"])
def test_exchange_rccl_one_rank(exe, self_rccl):
    """One-rank RCCL communicator: the rank's own block read in place (default: no transfer, no copy) and sent through
    ncclSend / ncclRecv to self (UPSP_EXCHANGE_SELF_RCCL=1: RCCL's own kernels carry it, as they carry every block between GPUs)."""
    r = subprocess.run([exe, "rccl1"], capture_output=True, text=True, timeout=300, env=dict(os.environ, UPSP_EXCHANGE_SELF_RCCL=self_rccl))
    assert r.returncode == 0, r.stdout + r.stderr
    assert r.stdout.count(" ok,") == 3, r.stdout


def test_exchange_rccl_two_ranks(exe, tmp_path):
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs (RCCL over xGMI)")
    idf = str(tmp_path / "nccl_id")
    ps = [subprocess.Popen([exe, "ranks", str(r), "2", idf, str(r)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
          for r in range(2)]
    outs = [p.communicate(timeout=300)[0] for p in ps]
    assert all(p.returncode == 0 for p in ps), outs
    assert all(o.count(" ok") == 8 and "FAILED" not in o for o in outs), outs


@pytest.mark.parametrize("world", [2, 3, 5])
def test_exchange_rank_processes_through_rccl_entry_points(exe, rccl_shim, tmp_path, world):
    """global_transpose and the reductions (cpp/exec/psp_process.cpp:707-771, 1866-1872) between `world` rank processes: the
    code path of an 8-GPU run -- upsp_comm_create from a shared id, one grouped send / receive per peer and chunk on the
    communicator's transfer stream, arrival marks, the owner's pass B on the received blocks -- with ragged node and frame
    shares (1003 nodes, 517 / 240 frames)."""
    idf = str(tmp_path / "nccl_id")
    env = dict(os.environ, UPSP_RCCL_LIBRARY=rccl_shim, UPSP_SHIM_TIMEOUT_S="60")
    ps = [subprocess.Popen([exe, "ranks", str(r), str(world), idf, "0"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, env=env)
          for r in range(world)]
    outs = [p.communicate(timeout=400)[0] for p in ps]
    assert all(p.returncode == 0 for p in ps), outs
    for o in outs:
        assert o.count(" ok") == 8 and "FAILED" not in o, o          # 3 row wires + 4 pixel shapes + two exchanges in turn
    if world > 1:
        sent = [int(l.split(", ")[1].split()[0]) for l in outs[0].splitlines() if " rows wire=" in l]
        assert sent[0] == 2 * sent[1] and 0 < sent[2] < sent[1]       # f32 : u16 : 12 bit really left the rank


def test_rccl_library_that_cannot_be_loaded_is_an_error(exe, tmp_path):
    """UPSP_RCCL_LIBRARY names the RCCL to bind; a path that does not load ends the run (no fallback to another copy)."""
    env = dict(os.environ, UPSP_RCCL_LIBRARY=str(tmp_path / "no_such_librccl.so"))
    r = subprocess.run([exe, "rccl1"], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode != 0 and "UPSP_RCCL_LIBRARY" in (r.stdout + r.stderr)
