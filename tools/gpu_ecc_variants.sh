# usage: bash tools/gpu_ecc_variants.sh "0 1 2"   -- bench --registration per UPSP_ECC_CVARIANT + kernel-trace of one sub-batch
set -o pipefail
out=gpurun_out/eccv; mkdir -p $out
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for v in $1; do
  UPSP_ECC_CVARIANT=$v timeout -k 10 300 python3 bench.py --registration --no-cpu-baseline > $out/reg_v$v.json 2> $out/reg_v$v.err || { echo "variant $v failed"; tail -3 $out/reg_v$v.err; continue; }
  UPSP_ECC_CVARIANT=$v timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace_v$v -- python3 tools/prof_ecc.py > $out/trace_v$v.log 2>&1
  f=$(find $out/trace_v$v -name "*kernel_stats.csv" | head -1)
  python3 - $out/reg_v$v.json "$f" $v <<'PY'
import json, sys, csv, re
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); k=d["kernels"]
print("variant", sys.argv[3], round(d["value"]), "frames/s step", round(d["ms_per_step"],2), "ecc", round(k["ecc_sums_kernel"]["ms_per_step"],2), k["ecc_sums_kernel"].get("launch_ms_min_median_max"))
if sys.argv[2]:
    for r in csv.DictReader(open(sys.argv[2])):
        m = re.search(r"\(anonymous namespace\)::(\w+)(<[^>]*>)?\(", r["Name"])
        if m and ("ecc_" in m.group(1) or "gauss" in m.group(1)):
            print("   %-44s calls %3s avg %8.1f us  max %8.1f us" % (m.group(1) + (m.group(2) or ""), r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MaxNs"]) / 1e3))
PY
  rm -rf $out/trace_v$v
done
