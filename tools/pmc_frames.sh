#!/bin/bash
# PMC passes (counters only) for the frame-loop kernels of the default bench step
mkdir -p gpurun_out/pmc_f
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INST_LEVEL_VMEM" "TCP_PENDING_STALL_CYCLES_sum TA_TA_BUSY_sum TCP_TCC_WRITE_REQ_sum TCP_TCC_READ_REQ_sum"; do
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d gpurun_out/pmc_f/p -- python3 bench.py --no-cpu-baseline --no-reraycast --steps 3 --warmup 1 > gpurun_out/pmc_f/p.log 2>&1
  f=$(find gpurun_out/pmc_f/p -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python3 - "$f" <<'PY'
import csv, sys, collections, re
agg = collections.OrderedDict()
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"]
    if "scan_compact" not in k and "node_rows" not in k: continue
    short = "scan_compact" if "scan_compact" in k else "node_rows"
    a = agg.setdefault((short, r["Counter_Name"]), [0, 0.0])
    a[0] += 1; a[1] += float(r["Counter_Value"])
for (k, c), (n, v) in agg.items(): print("%-14s %-34s calls=%d avg=%.5g" % (k, c, n, v / n))
PY
  rm -rf gpurun_out/pmc_f/p
done
