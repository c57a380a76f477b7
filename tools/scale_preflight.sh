#!/bin/bash
# Preflight for the first box with MORE THAN ONE GPU (this pool's boxes have one: real RCCL between two ranks has never run).
# From the repository root:      bash tools/scale_preflight.sh
#   1. the two tests that are skipped on one-GPU boxes: the exchange test program with rank processes over real RCCL / xGMI
#      (tests/test_exchange_gpu.py::test_exchange_rccl_two_ranks) and bench.py --gpus 2 (tests/test_bench_gpu.py::test_bench_two_ranks_rccl)
#   2. bench.py --gpus 2 --small with the system's RCCL (no UPSP_RCCL_LIBRARY), and which RCCL build the library bound
#   3. the same line's configs3 block at a reduced total (UPSP_BENCH_CONFIGS3_FRAMES=8192: 4096 frames per rank, four chunks)
# Stops at the first failing step.  Nothing here changes machine or GPU settings.
set -o pipefail
export HSA_ENABLE_IPC_MODE_LEGACY=0
unset UPSP_RCCL_LIBRARY UPSP_BENCH_BACKEND UPSP_BENCH_ONE_GPU
n=$(python3 -c 'import torch; print(torch.cuda.device_count())')
echo "preflight: $n GPU(s) visible"
if [ "$n" -lt 2 ]; then
  echo "preflight: needs two GPUs -- nothing run (on one GPU the same programs run through tests/shim, see tests/test_bench_gpu.py)"
  exit 0
fi
o=gpurun_out/preflight; mkdir -p $o
echo "== 1. the two-GPU tests"
timeout -k 10 900 python3 -m pytest tests/test_exchange_gpu.py::test_exchange_rccl_two_ranks tests/test_bench_gpu.py::test_bench_two_ranks_rccl -x -q -m gpu 2>&1 | tail -5 || exit 1
echo "== 2. bench.py --gpus 2 --small over RCCL"
UPSP_BENCH_CONFIGS3_FRAMES=8192 timeout -k 10 900 python3 bench.py --gpus 2 --small --steps 3 --warmup 1 > $o/line.json 2> $o/line.err || { tail -20 $o/line.err; exit 1; }
python3 - $o/line.json <<'PY' || exit 1
import json, sys
d = json.loads([l for l in open(sys.argv[1]).read().splitlines() if l.startswith("{")][-1])
print("value %.0f %s on %d GPUs, step %.3f ms (ranks min / max %s)" % (d["value"], d["unit"], d["n_gpus"], d["ms_per_step"], d.get("ms_per_step_rank_min_max")))
print("RCCL bound by the library:", d.get("rccl_bound"), "| ranks in its communicator:", d.get("rccl_nranks"))
print("exchange_self_check", d.get("exchange_self_check"), "| exchange_finals_check", d.get("exchange_finals_check"))
b = d.get("configs3") or {}
print("configs3 block:", {k: b.get(k) for k in ("frames_per_rank", "value", "ms_per_step", "rccl_nranks", "exchange_self_check", "exchange_finals_check", "note", "skipped")})
ok = d.get("rccl_nranks") == 2 and d.get("exchange_self_check") is True and "rccl_library" not in d and b.get("exchange_self_check") is True
print("preflight:", "OK" if ok else "FAILED")
sys.exit(0 if ok else 1)
PY
