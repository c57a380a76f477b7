set -o pipefail
python -m pytest tests -x -q -m gpu > gpurun_out/gpu_tests.log 2>&1; echo "tests rc=$?"
tail -4 gpurun_out/gpu_tests.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
bash tools/profile_bench.sh r02 2>&1 | tail -8
timeout -k 10 400 python3 bench.py --registration --frames 1000 > gpurun_out/prof_r02/bench_line_registration.json 2> gpurun_out/prof_r02/bench_line_registration.err; echo "reg rc=$?"
python3 - <<'PY'
import json
d=json.loads(open("gpurun_out/prof_r02/bench_line_registration.json").read().strip().splitlines()[-1])
print({k:d[k] for k in ("value","ms_per_step","ecc_iterations_per_frame")}); print(d["roofline"]); print(d.get("cpu_baseline"))
PY
timeout -k 10 300 python3 bench.py --force-chunked --no-cpu-baseline --no-reraycast > gpurun_out/prof_r02/bench_line_chunked.json 2>/dev/null; echo "chunked rc=$?"
timeout -k 10 300 python3 bench.py --fill-frame --no-cpu-baseline --no-reraycast > gpurun_out/prof_r02/bench_line_fill.json 2>/dev/null; echo "fill rc=$?"
timeout -k 10 300 python3 bench.py --overlap --no-cpu-baseline --no-reraycast > gpurun_out/prof_r02/bench_line_overlap.json 2>/dev/null; echo "overlap rc=$?"
python3 - <<'PY'
import json
for n in ("chunked","fill","overlap"):
    d=json.loads(open("gpurun_out/prof_r02/bench_line_%s.json" % n).read().strip().splitlines()[-1])
    print(n, {k:d[k] for k in ("value","ms_per_step","breakdown_ms")})
PY
