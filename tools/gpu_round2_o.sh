run() {
  timeout -k 10 200 python bench.py --no-cpu-baseline --no-reraycast --steps 6 --warmup 2 > gpurun_out/b_v.log 2>&1
  python - "$1" <<'PY'
import json, sys
d=json.loads(open("gpurun_out/b_v.log").read().strip().splitlines()[-1])
print("%-8s step %.3f ms" % (sys.argv[1], d["ms_per_step"]), {n:round(v["avg_launch_ms"]*1e3,1) for n,v in d["kernels"].items() if n in ("scan_compact_kernel","node_rows_kernel")})
PY
}
for i in 1 2 3 4 5 6 7 8; do run run$i; done
