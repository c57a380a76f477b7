# GPU suite + smoke + the driver's bench command (run from the repository root on the GPU box): what the driver does at round end
set -o pipefail
python -m pytest tests/ -x -q -m gpu > gpurun_out/full_tests.log 2>&1; echo rc=$? >> gpurun_out/full_tests.log; tail -4 gpurun_out/full_tests.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/full_bench.json 2> gpurun_out/full_bench.err; echo bench rc=$?
python - <<'PY'
import json
d=json.loads(open("gpurun_out/full_bench.json").read().strip().splitlines()[-1])
print(d["summary"]); print(d["ms_per_step"], d["parity_checked"], d["roofline"]["frac"], d["roofline"]["frac_of_measured"], d["roofline"]["step_frac_of_measured"])
print(d["configs2"]["value"], d["configs2"]["frames"], d["mrays_per_s"])
PY
