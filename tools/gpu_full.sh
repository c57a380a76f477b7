# whole GPU suite + smoke + default bench line (what the driver runs at round end)
set -o pipefail
out=gpurun_out/full; mkdir -p $out
python -m pytest tests -x -q -m gpu > $out/gpu_tests.log 2>&1; echo "tests rc=$?"; tail -5 $out/gpu_tests.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout -k 10 600 python3 bench.py > $out/bench_line.json 2> $out/bench_line.err; echo "bench rc=$?"; tail -2 $out/bench_line.err
python3 - <<'PY'
import json
d=json.loads(open("gpurun_out/full/bench_line.json").read().strip().splitlines()[-1])
print({k: d[k] for k in ("value","ms_per_step","breakdown_ms","parity_checked")}); print(d["roofline"])
c=d.get("configs2"); print(c and {k:c[k] for k in ("value","ms_per_step","ecc_iterations_per_frame")}, c and c["roofline"], c and c.get("cpu_baseline"))
print(d.get("parity"))
PY
