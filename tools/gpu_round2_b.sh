set -o pipefail
python -m pytest tests/test_bench_gpu.py tests/test_frames_gpu.py -x -q -m gpu > gpurun_out/t_bench.log 2>&1; echo "tests rc=$?"
tail -8 gpurun_out/t_bench.log
timeout -k 10 400 python bench.py > gpurun_out/bench_default.log 2>gpurun_out/bench_default.err; echo "bench rc=$?"
python - <<PY
import json
d=json.loads(open("gpurun_out/bench_default.log").read().strip().splitlines()[-1])
print({k:d[k] for k in ("value","ms_per_step","mrays_per_s","breakdown_ms","parity_checked","parity","reraycast_frames_per_s")})
print(d["roofline"]); print(d["cpu_baseline"])
print({n:(round(v["avg_launch_ms"]*1e3,1), v["calls_per_step"]) for n,v in d["kernels"].items()})
PY
timeout -k 10 300 python bench.py --force-chunked --no-cpu-baseline --no-reraycast > gpurun_out/bench_chunked.log 2>&1; echo "chunked rc=$?"
python - <<PY
import json
d=json.loads(open("gpurun_out/bench_chunked.log").read().strip().splitlines()[-1])
print({k:d[k] for k in ("value","ms_per_step","breakdown_ms")}, d["config"]["exchange"])
print({n:(round(v["avg_launch_ms"]*1e3,1), v["calls_per_step"]) for n,v in d["kernels"].items()})
PY
timeout -k 10 300 python bench.py --fill-frame --no-cpu-baseline --no-reraycast > gpurun_out/bench_fill.log 2>&1; echo "fill rc=$?"
python - <<PY
import json
d=json.loads(open("gpurun_out/bench_fill.log").read().strip().splitlines()[-1])
print({k:d[k] for k in ("value","ms_per_step","breakdown_ms")}, d["config"]["active_pixels"])
print({n:(round(v["avg_launch_ms"]*1e3,1), v["calls_per_step"]) for n,v in d["kernels"].items()})
PY
