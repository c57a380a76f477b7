# A/B of environment switches on the default bench (overlapped + serial), alternating: bash tools/gpu_ab_plain.sh "VAR=a" "VAR=b" ...
mkdir -p gpurun_out
for rep in 1 2; do
for e in "$@"; do
  for mode in "" "--serial"; do
  env $e timeout -k 10 300 python3 bench.py --no-cpu-baseline --no-reraycast $mode > gpurun_out/abp.json 2> gpurun_out/abp.err || { tail -3 gpurun_out/abp.err; continue; }
  python3 - "$e $mode" <<'PY'
import json,sys
d=json.loads(open("gpurun_out/abp.json").read().strip().splitlines()[-1]); k=d["kernels"]
g=lambda n: k.get(n, {"avg_launch_ms": 0.0})["avg_launch_ms"]
b=d["breakdown_ms"]
pr, pf = d.get("pixel_rays") or {}, d.get("pixel_rays_fill") or {}
print("%-24s %7.0f frames/s  step %.3f ms  build %.3f (alone %.3f) primary %.3f retry %.3f witness %.3f heavy %.3f  passA %.3f passB %.3f  pixel rays %.3f ms, fill %.3f ms" % (sys.argv[1], d["value"], d["ms_per_step"], b["projection_build"], b.get("projection_build_alone", 0), g("projection_kernel<primary>"), g("projection_kernel<retry>"), g("witness_kernels"), g("heavy_kernel"), g("scan_compact_kernel"), g("node_rows_kernel"), pr.get("ms", 0) or 0, pf.get("ms", 0) or 0), flush=True)
PY
  done
done
done
