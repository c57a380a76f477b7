#!/bin/bash
# Per-kernel durations of one registration sub-batch (rocprofv3 kernel trace of tools/prof_ecc.py); GPU box, repository root.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/eccprof
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/eccprof -- python3 tools/prof_ecc.py > gpurun_out/eccprof.log 2>&1
python3 - <<PY
import csv,glob
f=glob.glob("gpurun_out/eccprof/*/*kernel_stats.csv")[0]
for r in csv.DictReader(open(f)):
    if any(k in r["Name"] for k in ("ecc_", "gauss", "warp_", "node_rows", "hot_")):
        print("%-60s calls %4s avg %8.1f us  min %8.1f max %8.1f" % (r["Name"][:60], r["Calls"], float(r["AverageNs"])/1e3, float(r["MinNs"])/1e3, float(r["MaxNs"])/1e3))
PY
rm -rf gpurun_out/eccprof
