# PMC passes over tools/prof_ecc.py (counters only: --kernel-trace + --pmc, one group per run)
out=gpurun_out/pmc_ecc
mkdir -p $out
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
i=0
while read -r grp; do
  [ -z "$grp" ] && continue
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $out/p$i -- python3 tools/prof_ecc.py > $out/p$i.log 2>&1
  f=$(find $out/p$i -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python3 - "$f" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.OrderedDict()
for r in rows:
    k = r["Kernel_Name"]
    if "ecc_sums" not in k and "ecc_cols" not in k and "warp_u16" not in k and "gauss_fused" not in k: continue
    import re
    mm = re.search(r"\(anonymous namespace\)::(\w+)(<[^>]*>)?\(", k)
    short = ((mm.group(1) + (mm.group(2) or "")) if mm else k[:40]).replace("unsigned short", "u16")
    a = agg.setdefault((short, r["Counter_Name"]), [0, 0.0])
    a[0] += 1; a[1] += float(r["Counter_Value"])
for (k, c), (n, v) in agg.items():
    print("%-34s %-36s calls=%d avg=%.5g" % (k[:34], c, n, v / n))
PY
  rm -rf $out/p$i
done <<'GROUPS'
SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_VMEM_RD
SQ_ACTIVE_INST_VMEM SQ_INST_LEVEL_VMEM SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_THREAD_CYCLES_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAVES
TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum
TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_FLAT_READ_WAVEFRONTS_sum
GRBM_GUI_ACTIVE GRBM_TA_BUSY
GROUPS
