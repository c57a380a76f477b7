# A/B of environment switches on the registration bench, alternating: bash tools/gpu_ab.sh "VAR=a" "VAR=b" ...
mkdir -p gpurun_out
for rep in 1 2; do
for e in "$@"; do
  env $e timeout -k 10 300 python3 bench.py --registration --no-cpu-baseline --no-reraycast --steps 3 --warmup 1 > gpurun_out/ab.json 2> gpurun_out/ab.err || { tail -3 gpurun_out/ab.err; continue; }
  python3 - "$e" <<'PY'
import json,sys
d=json.loads(open("gpurun_out/ab.json").read().strip().splitlines()[-1]); k=d["kernels"]
g=lambda n: k.get(n, {"ms_per_step": 0.0})["ms_per_step"]
print("%-40s %7.0f frames/s  step %.2f ms  ecc %.2f (identity %.2f general %.2f)  gauss %.2f  solve %.2f" % (sys.argv[1], d["value"], d["ms_per_step"], g("ecc_sums_kernel"), g("ecc_sums_identity"), g("ecc_sums_general"), g("gauss_pass_kernels"), g("ecc_solve_kernel")), flush=True)
PY
done
done
