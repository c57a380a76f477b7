for e in "A=0" "UPSP_BENCH_BUILD_ON_SIDE=1 UPSP_BENCH_SIDE_PRIORITY=-1" "UPSP_BENCH_BUILD_ON_SIDE=1" "UPSP_BENCH_SIDE_PRIORITY=-1" "A=0" "UPSP_BENCH_BUILD_ON_SIDE=1 UPSP_BENCH_SIDE_PRIORITY=-1"; do
  env $e timeout -k 10 300 python3 bench.py --no-cpu-baseline --no-reraycast > gpurun_out/ov.json 2> gpurun_out/ov.err || { tail -2 gpurun_out/ov.err; continue; }
  python3 - "$e" <<'PY'
import json,sys
d=json.loads(open("gpurun_out/ov.json").read().strip().splitlines()[-1]); k=d["kernels"]
print("%-60s %7.0f frames/s  step %.3f ms  %s  A %.3f B %.3f primary %.3f" % (sys.argv[1], d["value"], d["ms_per_step"], {a:round(b,3) for a,b in d["breakdown_ms"].items()}, k["scan_compact_kernel"]["ms_per_step"], k["node_rows_kernel"]["ms_per_step"], k["projection_kernel<primary>"]["ms_per_step"]))
PY
done
