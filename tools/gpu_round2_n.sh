python -m pytest tests/test_frames_gpu.py -x -q -m gpu 2>&1 | tail -4
run() {
  timeout -k 10 300 python bench.py --no-reraycast --steps 10 --warmup 3 $2 > gpurun_out/b_v.log 2>gpurun_out/b_v.err; echo "rc=$?"
  python - "$1" <<'PY'
import json, sys
d=json.loads(open("gpurun_out/b_v.log").read().strip().splitlines()[-1])
print("%-10s fps %.0f step %.3f ms" % (sys.argv[1], d["value"], d["ms_per_step"]), d["breakdown_ms"], d.get("parity_checked"), {n:round(v["avg_launch_ms"]*1e3,1) for n,v in d["kernels"].items() if n in ("scan_compact_kernel","node_rows_kernel","projection_kernel<primary>","projection_kernel<retry>","witness_kernels")})
PY
}
run serial
run overlap --overlap
run serial
run overlap --overlap
