# A/B of environment switches on the 4-camera / 5 M-triangle bench: bash tools/gpu_multi_ab.sh "VAR=a" "VAR=b" ...
for e in "$@"; do
  env $e timeout -k 10 500 python3 bench.py --cameras 4 --model 5m --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/mab.json 2> gpurun_out/mab.err || { tail -3 gpurun_out/mab.err; continue; }
  python3 - "$e" <<'PY'
import json,sys
d=json.loads(open("gpurun_out/mab.json").read().strip().splitlines()[-1]); k=d["kernels"]
print("%-40s step %.2f ms  pass B %.3f ms (%.3f of peak)  pass A %.3f  parity %s" % (sys.argv[1], d["ms_per_step"], k["node_rows_multi_kernel"]["ms_per_step"], d["roofline"]["frac"], k["scan_compact_kernel"]["ms_per_step"], d.get("parity_checked")))
PY
done
