set -o pipefail
python -m pytest tests -x -q -m gpu > gpurun_out/gpu_tests.log 2>&1; echo "tests rc=$?"
tail -4 gpurun_out/gpu_tests.log
run() {
  timeout -k 10 200 python bench.py --no-cpu-baseline --no-reraycast --steps 10 --warmup 3 > gpurun_out/b_v.log 2>&1
  python - "$1" <<'PY'
import json, sys
d=json.loads(open("gpurun_out/b_v.log").read().strip().splitlines()[-1])
print("%-12s step %.3f ms proj %.3f" % (sys.argv[1], d["ms_per_step"], d["breakdown_ms"]["projection_build"]), {n:round(v["avg_launch_ms"]*1e3,1) for n,v in d["kernels"].items() if "projection" in n or "witness" in n or "prefetch" in n})
PY
}
run prefetch
UPSP_NO_PREFETCH=1 run noprefetch
run prefetch
UPSP_NO_PREFETCH=1 run noprefetch
timeout -k 10 400 env SOAK_SEED=40000 python tests/debug/soak_raycast.py 150 2>&1 | tail -4
