set -o pipefail
bash tools/profile_bench.sh r02 2>&1 | tail -45
echo ==== registration
timeout -k 10 400 python3 bench.py --registration --frames 1000 > gpurun_out/prof_r02/bench_line_registration.json 2> gpurun_out/prof_r02/bench_line_registration.err; echo "rc=$?"
python3 - <<'PY'
import json
d=json.loads(open("gpurun_out/prof_r02/bench_line_registration.json").read().strip().splitlines()[-1])
print({k:d[k] for k in ("value","ms_per_step","ecc_iterations_per_frame")}); print(d["roofline"]); print(d.get("cpu_baseline"))
print({n:(round(v["ms_per_step"],3), round(v["avg_launch_ms"]*1e3,1)) for n,v in d["kernels"].items()})
PY
echo ==== chunked
timeout -k 10 300 python3 bench.py --force-chunked --no-cpu-baseline --no-reraycast > gpurun_out/prof_r02/bench_line_chunked.json 2>/dev/null; echo "rc=$?"
python3 - <<'PY'
import json
d=json.loads(open("gpurun_out/prof_r02/bench_line_chunked.json").read().strip().splitlines()[-1])
print({k:d[k] for k in ("value","ms_per_step","breakdown_ms")}, d["config"]["exchange"])
PY
echo ==== fill-frame
timeout -k 10 300 python3 bench.py --fill-frame --no-cpu-baseline --no-reraycast > gpurun_out/prof_r02/bench_line_fill.json 2>/dev/null; echo "rc=$?"
python3 - <<'PY'
import json
d=json.loads(open("gpurun_out/prof_r02/bench_line_fill.json").read().strip().splitlines()[-1])
print({k:d[k] for k in ("value","ms_per_step","breakdown_ms")}, d["config"]["active_pixels"])
print({n:(round(v["avg_launch_ms"]*1e3,1)) for n,v in d["kernels"].items()})
PY
echo ==== 2 ranks gloo one gpu
UPSP_BENCH_BACKEND=gloo UPSP_BENCH_ONE_GPU=1 timeout -k 10 300 python3 bench.py --gpus 2 --no-cpu-baseline --no-reraycast > gpurun_out/prof_r02/bench_line_2ranks_gloo.json 2>/dev/null; echo "rc=$?"
python3 - <<'PY'
import json
d=json.loads(open("gpurun_out/prof_r02/bench_line_2ranks_gloo.json").read().strip().splitlines()[-1])
print({k:d[k] for k in ("value","n_gpus","ms_per_step","breakdown_ms")}, d["config"]["exchange"])
PY
ls -la gpurun_out/prof_r02
