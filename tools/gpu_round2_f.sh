run() {
  timeout -k 10 200 python bench.py --no-cpu-baseline --no-reraycast --steps 5 --warmup 2 $2 > gpurun_out/b_v.log 2>&1
  python - "$1" <<'PY'
import json, sys
d=json.loads(open("gpurun_out/b_v.log").read().strip().splitlines()[-1])
print("%-24s step %.3f ms frame_loop %.3f" % (sys.argv[1], d["ms_per_step"], d["breakdown_ms"]["frame_loop"]), {n:round(v["avg_launch_ms"]*1e3,1) for n,v in d["kernels"].items() if n in ("scan_compact_kernel","node_rows_kernel","projection_kernel<primary>")}, d["config"]["active_pixels"])
PY
}
run base
UPSP_DBG_NOORDER=1 run noorder
run base-plain --plain-frames
UPSP_DBG_NOORDER=1 run noorder-plain --plain-frames
timeout -k 10 100 tools/probe/passA_stages c 2>&1 | grep "stage 3 nt\|stage 0 nt"
timeout -k 10 100 tools/probe/passA_stages 2>&1 | grep "stage 3 nt\|stage 0 nt"
