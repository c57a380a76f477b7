#!/bin/bash
# PMC passes for the traversal kernels (counters only: --kernel-trace + --pmc, one group per run)
# usage: tools/pmc_passes.sh <outdir> -- <python script args...>
out=$1; shift; shift
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
i=0
while read -r grp; do
  [ -z "$grp" ] && continue
  i=$((i+1))
  timeout 600 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $out/p$i -- python "$@" > $out.p$i.log 2>&1
  f=$(find $out/p$i -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python3 - "$f" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.OrderedDict()
for r in rows:
    k = r["Kernel_Name"]
    if "upsp" not in k: continue
    short = k.split("(anonymous namespace)::")[-1].split("(")[0]
    key = (short, r["Counter_Name"])
    a = agg.setdefault(key, [0, 0.0, 0.0])
    a[0] += 1; a[1] += float(r["Counter_Value"]); a[2] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
for (k, c), (n, v, t) in agg.items():
    print("%-28s %-36s calls=%d avg=%.4g avg_ns=%.0f" % (k[:28], c, n, v / n, t / n))
PY
done <<'GROUPS'
SQ_WAIT_ANY SQ_ACTIVE_INST_VMEM SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM_RD SQ_THREAD_CYCLES_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_SCA
TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum
TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_REQUEST_sum TCP_TCP_LATENCY_sum TCP_TA_TCP_STATE_READ_sum
TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_REQ_sum
TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_FLAT_READ_WAVEFRONTS_sum
GRBM_GUI_ACTIVE GRBM_TA_BUSY
GROUPS
