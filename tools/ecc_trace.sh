#!/bin/bash
# Per-launch durations of the ECC kernels of one registration step (rocprofv3 kernel trace); GPU box, repository root.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/eccprof -- python3 bench.py --registration --no-cpu-baseline --no-reraycast --steps 1 --warmup 1 > gpurun_out/eccprof.log 2>&1
python3 - <<PY
import csv,glob,re
f=glob.glob("gpurun_out/eccprof/*/*kernel_trace.csv")[0]
rows=[r for r in csv.DictReader(open(f)) if "ecc_" in r["Kernel_Name"] and ("sums" in r["Kernel_Name"] or "interior" in r["Kernel_Name"] or "border" in r["Kernel_Name"])]
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
def tag(n):
    m=re.search(r"ecc_(\w+?)_kernel(<(\w+)>)?", n); return (m.group(1)[0].upper()+("t" if m.group(3)=="true" else "f" if m.group(3) else ""))
print(" ".join("%s%.0f" % (tag(r["Kernel_Name"]), (int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3) for r in rows[-60:]))
PY
rm -rf gpurun_out/eccprof
