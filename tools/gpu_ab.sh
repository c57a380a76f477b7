# A/B of one environment switch on the registration bench: bash tools/gpu_ab.sh "VAR=a" "VAR=b" ...
for e in "$@"; do
  env $e timeout -k 10 300 python3 bench.py --registration --no-cpu-baseline --no-reraycast --steps 3 --warmup 1 > gpurun_out/ab.json 2> gpurun_out/ab.err || { tail -3 gpurun_out/ab.err; continue; }
  python3 - "$e" <<'PY'
import json,sys
d=json.loads(open("gpurun_out/ab.json").read().strip().splitlines()[-1]); k=d["kernels"]
print("%-40s %7.0f frames/s  step %.2f ms  ecc %.2f  gauss %.2f  solve %.2f" % (sys.argv[1], d["value"], d["ms_per_step"], k["ecc_sums_kernel"]["ms_per_step"], k["gauss_pass_kernels"]["ms_per_step"], k.get("ecc_solve_kernel", {"ms_per_step": 0.0})["ms_per_step"]))
PY
done
