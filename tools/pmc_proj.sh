out=gpurun_out/pmc_proj
mkdir -p $out
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
i=0
while read -r grp; do
  [ -z "$grp" ] && continue
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $out/p$i -- python3 tools/prof_proj.py > $out/p$i.log 2>&1
  f=$(find $out/p$i -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python3 - "$f" <<'PY'
import csv, sys, collections, re
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.OrderedDict()
for r in rows:
    k = r["Kernel_Name"]
    if "projection_kernel" not in k and "witness_kernel" not in k: continue
    m = re.search(r"(projection_kernel<[^>]*>|witness_kernel)", k)
    a = agg.setdefault((m.group(1), r["Counter_Name"]), [0, 0.0])
    a[0] += 1; a[1] += float(r["Counter_Value"])
for (k, c), (n, v) in agg.items():
    print("%-30s %-28s calls=%d avg=%.5g" % (k, c, n, v / n))
PY
  rm -rf $out/p$i
done <<'GROUPS'
SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU
SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INST_LEVEL_VMEM SQ_WAVES SQ_ACTIVE_INST_MISC
GRBM_GUI_ACTIVE
GROUPS
