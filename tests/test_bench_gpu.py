"""GPU: bench.py contract -- the JSON line, the in-run parity check against the oracle, and the
N > 1 launch path (`--gpus N` starts its own ranks).  Sub-processes: the test process keeps its own
GPU context, every bench run is a fresh child."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_bench(args, env=None, timeout=600):
    e = dict(os.environ)
    e.pop("RANK", None); e.pop("WORLD_SIZE", None); e.pop("LOCAL_RANK", None)
    e.update(env or {})
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, cwd=ROOT, env=e,
                       capture_output=True, text=True, timeout=timeout)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    return json.loads(lines[0])


def test_bench_small_line_and_parity(gpu_lib):
    d = run_bench(["--small", "--steps", "2", "--warmup", "1"])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
              "scaling", "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["metric"] == "frames/s"
    assert d["parity_checked"] is True and all(d["parity"].values()), d["parity"]
    r = d["roofline"]
    assert r["bound"] == "hbm" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
    assert r["traffic"] is None                        # no PMC summary was handed over
    # the measured denominators: the read-only / write-only probe of the kernel's own stream -- a fraction above 1 would mean the
    # probe is slower than the product kernel it normalises (round-5 review: the copy probe was)
    assert 0.0 < r["frac_of_measured"] <= 1.0 and 0.0 < r["step_frac_of_measured"] <= 1.0, r
    assert "read-only" in r["peak_measured_kind"] and "write-only" in r["peak_measured_kind"]
    assert d["parity"]["step_call_same_projection"] is True     # the default step = ONE upsp_pipeline_step call, checked through it
    assert d["cpu_baseline"]["kind"] == "port" and d["cpu_baseline"]["cores"] >= 1


def test_bench_small_registration_parity(gpu_lib):
    """`--registration` (configs[2] shape) checks itself too: warps, ECC iteration counts and registered rows of the
    frames the CPU baseline registered, against the oracle, at the bench's own 1024 x 1024 image size."""
    d = run_bench(["--small", "--registration", "--steps", "1", "--warmup", "1"])
    assert d["config"]["workload"].startswith("configs[2]")
    assert d["parity_checked"] is True and all(d["parity"].values()), d["parity"]
    for k in ("registration_ecc_warp_linear_1e-4", "registration_ecc_warp_translation_2e-3_px",
              "registration_ecc_iteration_counts", "registration_rows_bitexact_for_gpu_warp"):
        assert d["parity"][k] is True, k
    assert d["registration_parity"]["frames"] >= 2 and d["ecc_iterations_per_frame"] >= 1.0
    assert d["roofline"]["kernel"] in d["kernels"]


def test_bench_multi_camera_line(gpu_lib):
    """`--cameras 2` (configs[4] shape, reduced): the weighted multi-camera loop produces its line and checks itself."""
    d = run_bench(["--small", "--cameras", "2", "--size", "512", "--steps", "1", "--warmup", "1"])
    assert d["config"]["workload"].startswith("configs[4]") and d["n_gpus"] == 1
    assert d["parity_checked"] is True and all(d["parity"].values()), d["parity"]
    assert d["roofline"]["bound"] == "hbm" and d["cpu_baseline"]["kind"] == "port"


def test_bench_two_ranks_on_one_gpu(gpu_lib, rccl_shim):
    """`--gpus 2` launches its own two ranks; on a one-GPU box both sit on cuda:0, torch.distributed's rendezvous runs over gloo and
    the library's exchange (the RCCL branch of csrc/exchange.hip) through the tests' stand-in RCCL.  Packed u16 rows in 4 chunks."""
    from conftest import one_gpu_ranks_env
    d = run_bench(["--gpus", "2", "--row-wire", "--small", "--steps", "2", "--warmup", "1"], env=one_gpu_ranks_env(rccl_shim))
    assert d["n_gpus"] == 2 and d["exchange_self_check"] is True and d["rccl_nranks"] == 2
    assert d["exchange_finals_check"] in (True, None)       # (True with the pixel-series wire: finals of the all-reduced sums)
    assert d["exchange_bytes_per_step"]["transport"].startswith("C ABI") and d["exchange_bytes_per_step"]["sent_to_other_ranks_this_run"] > 0
    assert d["config"]["exchange"] == "4 chunks, visible rows as u16"
    assert d["config"]["parallelism"] == "frames sharded x2"


def test_bench_rccl_one_rank_group(gpu_lib):
    """RCCL refuses two ranks on one GPU, so on a one-GPU box the N > 1 loop runs in a ONE-rank nccl group with the
    collectives forced (all_reduce of the accumulators, async all_to_all_single of the packed u16 chunks as bytes,
    barrier): the calls, dtypes and split sizes go through RCCL itself."""
    # (UPSP_EXCHANGE_SELF_RCCL=1: the rank's own block goes through ncclSend / ncclRecv to self as well -- by default a rank reads its
    #  own block where it lies in the send buffer, and a one-rank group would then issue no point-to-point call at all)
    d = run_bench(["--force-chunked", "--row-wire", "--small", "--steps", "2", "--warmup", "1"],
                  env={"UPSP_FORCE_COLLECTIVES": "1", "UPSP_EXCHANGE_SELF_RCCL": "1"})
    assert d["n_gpus"] == 1 and d["collectives"].startswith("issued through RCCL")
    assert d["config"]["exchange"] == "4 chunks, visible rows as u16"
    assert d["parity_checked"] is True


def test_bench_pixel_wire(gpu_lib, rccl_shim):
    """The default of the N > 1 loop: the active pixels' series travel and the owner of a node runs pass B over all frames -- through the
    C-ABI exchange over RCCL (one-rank group) and between two rank processes on this GPU (stand-in RCCL); the series that come
    out are compared with the oracle in the run (one rank) like the row exchange's."""
    from conftest import one_gpu_ranks_env
    two = one_gpu_ranks_env(rccl_shim)
    d = run_bench(["--force-chunked", "--small", "--steps", "2", "--warmup", "1"], env={"UPSP_FORCE_COLLECTIVES": "1"})
    assert d["config"]["exchange"] == "1 chunks, active-pixel series as u16; pass A once for the rank's frames beside the projection build"
    assert d["parity_checked"] is True
    x = d["exchange_bytes_per_step"]
    assert x["what_travels"] == "active-pixel series" and x["transport"].startswith("C ABI")
    d2 = run_bench(["--force-chunked", "--row-wire", "--small", "--steps", "1", "--warmup", "1", "--no-cpu-baseline", "--no-reraycast"],
                   env={"UPSP_FORCE_COLLECTIVES": "1"})
    assert x["travelling_rows"] < d2["exchange_bytes_per_step"]["travelling_rows"]          # fewer pixel rows than node rows
    # two ranks: two exchanges in turn (a step's series are finished behind the next step's chunks; the last one before the
    # clock stops) -- and the same schedule forced on one rank through RCCL, parity checked against the oracle in the run
    d = run_bench(["--gpus", "2", "--small", "--steps", "3", "--warmup", "1"], env=two)
    assert d["rccl_nranks"] == 2 and d["exchange_bytes_per_step"]["sent_to_other_ranks_this_run"] > 0
    # (deferred = the N > 1 default: one block per peer, pass A once for the rank's frames beside the projection build)
    once = "1 chunks, active-pixel series as u16; two exchanges in turn, a step's series finished behind the next step's chunks; pass A once"
    assert d["n_gpus"] == 2 and d["config"]["exchange"].startswith(once) and d["exchange_self_check"] is True
    d = run_bench(["--force-chunked", "--defer-exchange", "--small", "--steps", "3", "--warmup", "1"],
                  env={"UPSP_FORCE_COLLECTIVES": "1", "UPSP_EXCHANGE_SELF_RCCL": "1"})
    assert d["config"]["exchange"].startswith(once) and d["parity_checked"] is True
    assert d["rccl_nranks"] == 1
    # the schedules of round 3 / early round 4 stay reachable: deferred with pass A per chunk, and finished inside the step
    d = run_bench(["--force-chunked", "--defer-exchange", "--chunk-scan", "--small", "--steps", "3", "--warmup", "1"], env={"UPSP_FORCE_COLLECTIVES": "1"})
    assert d["config"]["exchange"].startswith("4 chunks, active-pixel series as u16; two exchanges in turn") and d["parity_checked"] is True
    d = run_bench(["--force-chunked", "--sync-exchange", "--chunks", "3", "--small", "--steps", "2", "--warmup", "1"], env={"UPSP_FORCE_COLLECTIVES": "1"})
    assert d["config"]["exchange"] == "3 chunks, active-pixel series as u16; pass A once for the rank's frames beside the projection build"
    assert d["parity_checked"] is True
    d = run_bench(["--force-chunked", "--sync-exchange", "--chunk-scan", "--small", "--steps", "2", "--warmup", "1"], env={"UPSP_FORCE_COLLECTIVES": "1"})
    assert d["config"]["exchange"] == "4 chunks, active-pixel series as u16; pass A per chunk after the projection build"
    assert d["parity_checked"] is True
    d = run_bench(["--gpus", "2", "--small", "--steps", "2", "--warmup", "1", "--sync-exchange"], env=two)
    assert d["config"]["exchange"] == "1 chunks, active-pixel series as u16; pass A once for the rank's frames beside the projection build"


def test_bench_two_ranks_rccl(gpu_lib):
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs (RCCL over xGMI)")
    d = run_bench(["--gpus", "2", "--small", "--steps", "2", "--warmup", "1"])
    assert d["n_gpus"] == 2 and "backend" not in d and d["rccl_nranks"] == 2
    assert d["config"]["exchange"].startswith("1 chunks, active-pixel series as u16; two exchanges in turn")


# (at most 4 ranks: the pool allows six processes with the GPU open -- this test process, torchrun's launcher and the ranks)
@pytest.mark.parametrize("world,extra", [(4, []), (3, ["--sync-exchange"]), (3, ["--chunk-scan"]), (4, ["--row-wire", "--wire12"])])
def test_bench_rank_processes_on_one_gpu(gpu_lib, rccl_shim, world, extra):
    """`bench.py --gpus N` semantics with N rank processes on this GPU: the N > 1 loop end to end through the library's exchange
    (upsp_comm_create from a broadcast id, grouped sends / receives per peer, owner's pass B, all-reduce of the sums) -- deferred
    and in-step schedules, pass A once / per chunk, packed 12-bit rows -- every rank checks its series slice against its own
    frames (exchange_self_check, all-reduced).  Needs one GPU only; on a multi-GPU box the ranks still share cuda:0 here."""
    env = {"UPSP_BENCH_BACKEND": "gloo", "UPSP_BENCH_ONE_GPU": "1", "UPSP_RCCL_LIBRARY": rccl_shim}
    d = run_bench(["--gpus", str(world), "--small", "--steps", "2", "--warmup", "1"] + extra, env=env)
    assert d["n_gpus"] == world and d["rccl_nranks"] == world and d["exchange_self_check"] is True
    assert d["exchange_finals_check"] in (True, None)
    x = d["exchange_bytes_per_step"]
    assert x["transport"].startswith("C ABI") and x["sent_to_other_ranks_this_run"] > 0
    assert d["config"]["parallelism"] == "frames sharded x%d" % world


def test_bench_configs3_block_in_the_multi_rank_line(gpu_lib, rccl_shim):
    """The N > 1 line carries BASELINE configs[3] as a block of its own: the run's total frames (here 6 600 instead of 100 000)
    sharded over the ranks, every share resident, ONE step through the chunked pixel-series exchange with real peer processes
    (2 200 frames per rank = three pass-A chunks), with its own value, per-rank step times, communicator size, self / finals checks
    and exchange bytes -- the line a multi-GPU run prints must be right the first time (round-5 review)."""
    env = {"UPSP_BENCH_BACKEND": "gloo", "UPSP_BENCH_ONE_GPU": "1", "UPSP_RCCL_LIBRARY": rccl_shim, "UPSP_BENCH_CONFIGS3_FRAMES": "6600"}
    d = run_bench(["--gpus", "3", "--small", "--steps", "2", "--warmup", "1"], env=env, timeout=900)
    assert d["n_gpus"] == 3 and d["exchange_self_check"] is True
    b = d["configs3"]
    assert b["frames_per_rank"] == 2200 and b["workload"].startswith("configs[3]: 6600 frames")
    assert b["rccl_nranks"] == 3 and b["exchange_self_check"] is True and b["exchange_finals_check"] in (True, None)
    assert b["value"] > 0 and b["unit"] == "frames/s" and len(b["ms_per_step_rank_min_max"]) == 2
    assert b["exchange_bytes_per_step"]["frames_per_rank"] == 2200 and b["exchange_bytes_per_step"]["sent_to_other_ranks_this_run"] > 0
    assert b["exchange"].startswith("3 chunks, active-pixel series")
