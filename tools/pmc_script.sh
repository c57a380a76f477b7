#!/bin/bash
# Counter passes (rocprofv3 --kernel-trace --pmc, one group per run -- never combined with other trace domains) over a
# python script; per-kernel averages of the FULL-SIZE launches (launches shorter than a third of the longest of their
# kernel are left out) for the kernels whose short name matches the filter.
#   usage (GPU box, repository root):  bash tools/pmc_script.sh "<kernel regex>" tools/prof_xxx.py [args] > profiles/r03_xxx_pmc.txt
# The program sits directly after `--` (python3 itself; no env / bash -c hop).
filt=$1; shift
out=gpurun_out/pmc_tmp_$$
mkdir -p $out
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
echo "# rocprofv3 --kernel-trace --pmc <group> -- python3 $*   (kernels matching /$filt/; FETCH_SIZE / WRITE_SIZE in KB)"
i=0
while read -r grp; do
  [ -z "$grp" ] && continue
  i=$((i+1))
  timeout -k 10 ${PMC_TIMEOUT:-240} rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $out/p$i -- python3 "$@" > $out/p$i.log 2>&1
  f=$(find $out/p$i -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python3 - "$f" "$filt" <<'PY'
import csv, sys, collections, re
rows = list(csv.DictReader(open(sys.argv[1])))
pat = re.compile(sys.argv[2])
def short(k):
    mm = re.search(r"\(anonymous namespace\)::(\w+)(<[^>]*>)?\(", k)
    return ((mm.group(1) + (mm.group(2) or "")) if mm else k[:40]).replace("unsigned short", "u16")
dur = collections.defaultdict(float)
for r in rows:
    k = short(r["Kernel_Name"])
    if pat.search(k): dur[k] = max(dur[k], int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
agg = collections.OrderedDict()
for r in rows:
    k = short(r["Kernel_Name"])
    if not pat.search(k): continue
    d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    if d * 3 < dur[k]: continue
    a = agg.setdefault((k, r["Counter_Name"]), [0, 0.0, 0.0])
    a[0] += 1; a[1] += float(r["Counter_Value"]); a[2] += d
for (k, c), (n, v, t) in agg.items():
    print("%-40s %-34s launches=%-3d avg=%-12.6g avg_us=%.1f" % (k[:40], c, n, v / n, t / n / 1e3))
PY
done <<'GROUPS'
SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_VMEM_RD
SQ_INST_LEVEL_VMEM SQ_INSTS_SALU SQ_THREAD_CYCLES_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAVES SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_SCA
TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum
TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum
FETCH_SIZE
WRITE_SIZE
GRBM_GUI_ACTIVE GRBM_TA_BUSY
SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SMEM SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM
GROUPS
rm -rf $out
