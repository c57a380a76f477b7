set -o pipefail
mkdir -p gpurun_out/r3f
python -m pytest tests/test_frames_gpu.py tests/test_configs_gpu.py tests/test_psp_gpu.py -x -q -m gpu > gpurun_out/r3f/tests.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -4 gpurun_out/r3f/tests.log
[ $rc -eq 0 ] || exit 1
timeout -k 10 500 python3 bench.py --cameras 4 --model 5m --steps 3 --warmup 1 > gpurun_out/r3f/multi.json 2> gpurun_out/r3f/multi.err; echo "multi rc=$?"; tail -3 gpurun_out/r3f/multi.err
for r in 2 8; do
UPSP_MULTI_ROWS=$r timeout -k 10 300 python3 bench.py --cameras 4 --model 5m --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r3f/multi_rows$r.json 2> gpurun_out/r3f/multi_rows$r.err; echo "rows $r rc=$?"
done
python3 - <<'PY'
import json
for n in ("multi","multi_rows2","multi_rows8"):
    try:
        d=json.loads(open("gpurun_out/r3f/%s.json" % n).read().strip().splitlines()[-1])
        print(n, round(d["value"]), "frame sets/s", round(d["ms_per_step"],2), d["breakdown_ms"], "rowGB/s", round(d["pass_b_row_GBps"]), d["roofline"]["frac"], d.get("parity"), d.get("cpu_baseline"))
        for k,v in d["kernels"].items(): print("   ", k, round(v["ms_per_step"],3), v.get("achieved_GBps"))
    except Exception as e: print(n, "failed", e)
PY
