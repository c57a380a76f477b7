# single-camera pass B: series loads one sweep ahead (default) against loads at the point of use, --serial so that nothing runs beside
for e in "UPSP_ROWS_AHEAD=1" "UPSP_ROWS_AHEAD=0" "UPSP_ROWS_AHEAD=1" "UPSP_ROWS_AHEAD=0"; do
  env $e timeout -k 10 300 python3 bench.py --serial --no-cpu-baseline --no-reraycast > gpurun_out/ov.json 2> gpurun_out/ov.err || { tail -3 gpurun_out/ov.err; continue; }
  python3 - "$e" <<'PY'
import json,sys
d=json.loads(open("gpurun_out/ov.json").read().strip().splitlines()[-1]); k=d["kernels"]
print("%-20s %7.0f frames/s  step %.3f ms  A %.3f B %.3f %s" % (sys.argv[1], d["value"], d["ms_per_step"], k["scan_compact_kernel"]["ms_per_step"], k["node_rows_kernel"]["ms_per_step"], k["node_rows_kernel"].get("launch_ms_min_median_max")))
PY
done
