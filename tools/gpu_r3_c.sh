set -o pipefail
mkdir -p gpurun_out/r3c
python -m pytest tests/test_imageops_gpu.py tests/test_psp_gpu.py -x -q -m gpu -s > gpurun_out/r3c/tests.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -6 gpurun_out/r3c/tests.log
[ $rc -eq 0 ] || exit 1
for v in 0 1 2 3; do
  UPSP_ECC_CVARIANT=$v timeout -k 10 300 python3 bench.py --registration --no-cpu-baseline > gpurun_out/r3c/reg_v$v.json 2> gpurun_out/r3c/reg_v$v.err; echo "variant $v rc=$?"
done
UPSP_ECC_KERNEL=2 timeout -k 10 300 python3 bench.py --registration --no-cpu-baseline > gpurun_out/r3c/reg_old.json 2> gpurun_out/r3c/reg_old.err; echo "old rc=$?"
python3 - <<'PY'
import json
for n in ("v0","v1","v2","v3","old"):
    try:
        d=json.loads(open("gpurun_out/r3c/reg_%s.json" % n).read().strip().splitlines()[-1])
        k=d["kernels"]
        print(n, round(d["value"]), round(d["ms_per_step"],2), "ecc", round(k["ecc_sums_kernel"]["ms_per_step"],2), k["ecc_sums_kernel"].get("launch_ms_min_median_max"), "solve", round(k["ecc_solve_kernel"]["ms_per_step"],2), "its", d["ecc_iterations_per_frame"])
    except Exception as e: print(n, "failed", e)
PY
