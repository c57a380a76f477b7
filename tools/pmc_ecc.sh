# PMC passes over tools/prof_ecc.py (counters only: --kernel-trace + --pmc, one group per run); prints per-kernel
# averages over the FULL-SIZE launches (launches shorter than a third of the longest of their kernel are left out:
# late iterations with a few frames).   bash tools/pmc_ecc.sh > gpurun_out/ecc_pmc.txt
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
import csv, sys, collections, re
rows = list(csv.DictReader(open(sys.argv[1])))
def short(k):
    mm = re.search(r"\(anonymous namespace\)::(\w+)(<[^>]*>)?\(", k)
    return ((mm.group(1) + (mm.group(2) or "")) if mm else k[:40]).replace("unsigned short", "u16")
want = ("ecc_sums", "ecc_cols", "gauss_fused", "ecc_solve", "gauss5_")
dur = collections.defaultdict(float)
for r in rows:
    k = short(r["Kernel_Name"])
    if any(w in k for w in want):
        dur[k] = max(dur[k], int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
agg = collections.OrderedDict()
for r in rows:
    k = short(r["Kernel_Name"])
    if not any(w in k for w in want): continue
    d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    if d * 3 < dur[k]: continue
    a = agg.setdefault((k, r["Counter_Name"]), [0, 0.0, 0.0])
    a[0] += 1; a[1] += float(r["Counter_Value"]); a[2] += d
for (k, c), (n, v, t) in agg.items():
    print("%-36s %-36s launches=%d avg=%.5g  (avg %.1f us under the counters)" % (k[:36], c, n, v / n, t / n / 1e3))
PY
  rm -rf $out/p$i
done <<'GROUPS'
SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_VMEM_RD
SQ_ACTIVE_INST_VMEM SQ_INST_LEVEL_VMEM SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_THREAD_CYCLES_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAVES
TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum
TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_FLAT_READ_WAVEFRONTS_sum
TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum
GRBM_GUI_ACTIVE GRBM_TA_BUSY
SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS
GROUPS
