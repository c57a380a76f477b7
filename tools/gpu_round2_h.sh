run() {
  timeout -k 10 200 python bench.py --no-cpu-baseline --no-reraycast --steps 10 --warmup 3 > gpurun_out/b_v.log 2>&1
  python - "$1" <<'PY'
import json, sys
d=json.loads(open("gpurun_out/b_v.log").read().strip().splitlines()[-1])
print("%-12s step %.3f ms proj %.3f" % (sys.argv[1], d["ms_per_step"], d["breakdown_ms"]["projection_build"]), {n:round(v["avg_launch_ms"]*1e3,1) for n,v in d["kernels"].items() if "projection" in n or "witness" in n})
PY
}
run xcd
UPSP_XCD_AWARE=0 run noxcd
run xcd
UPSP_XCD_AWARE=0 run noxcd
python -m pytest tests/test_raycast_gpu.py tests/test_projection_gpu.py -x -q -m gpu 2>&1 | tail -3
