set -o pipefail
mkdir -p gpurun_out/prof_r03
mkdir -p gpurun_out/r3f
python -m pytest tests/test_frames_gpu.py tests/test_configs_gpu.py tests/test_psp_gpu.py -x -q -m gpu > gpurun_out/r3f/tests.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -3 gpurun_out/r3f/tests.log
[ $rc -eq 0 ] || exit 1
timeout -k 10 300 python3 bench.py --cameras 4 --model 5m --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r3f/multi.json 2> gpurun_out/r3f/multi.err; echo "multi rc=$?"
python3 - <<'PY'
import json
d=json.loads(open("gpurun_out/r3f/multi.json").read().strip().splitlines()[-1])
print(round(d["value"]), "frame sets/s", round(d["ms_per_step"],2), "rowGB/s", round(d["pass_b_row_GBps"]), d["roofline"]["frac"], d["kernels"]["node_rows_multi_kernel"])
PY
