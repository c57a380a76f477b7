set -o pipefail
mkdir -p gpurun_out/r3d
python -m pytest tests/test_imageops_gpu.py tests/test_psp_gpu.py -x -q -m gpu -s > gpurun_out/r3d/tests.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -4 gpurun_out/r3d/tests.log
[ $rc -eq 0 ] || exit 1
for v in 0 3; do
  UPSP_ECC_CVARIANT=$v timeout -k 10 300 python3 bench.py --registration --no-cpu-baseline > gpurun_out/r3d/reg_v$v.json 2> gpurun_out/r3d/reg_v$v.err; echo "variant $v rc=$?"
done
python3 - <<'PY'
import json
for n in ("v0","v3"):
    try:
        d=json.loads(open("gpurun_out/r3d/reg_%s.json" % n).read().strip().splitlines()[-1])
        k=d["kernels"]
        print(n, round(d["value"]), round(d["ms_per_step"],2), "ecc", round(k["ecc_sums_kernel"]["ms_per_step"],2), k["ecc_sums_kernel"].get("launch_ms_min_median_max"), "solve", round(k["ecc_solve_kernel"]["ms_per_step"],2), "its", d["ecc_iterations_per_frame"])
    except Exception as e: print(n, "failed", e)
PY
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r3d/trace -- python3 tools/prof_ecc.py > gpurun_out/r3d/trace.log 2>&1; echo "trace rc=$?"
f=$(find gpurun_out/r3d/trace -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f gpurun_out/r3d/kernel_stats.csv && head -12 gpurun_out/r3d/kernel_stats.csv | cut -c1-200
rm -rf gpurun_out/r3d/trace
bash tools/pmc_ecc.sh > gpurun_out/r3d/ecc_pmc.txt 2>&1; echo "pmc rc=$?"; cat gpurun_out/r3d/ecc_pmc.txt | grep -v "warp_u16" | head -90
