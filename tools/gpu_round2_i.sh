cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/prof_ecc
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_ecc -- python3 tools/prof_ecc.py > gpurun_out/prof_ecc.log 2>&1
f=$(find gpurun_out/prof_ecc -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "upsp" in r["Name"]]
rows.sort(key=lambda r: -int(r["TotalDurationNs"]))
for r in rows[:12]:
    print("%-100s calls=%5s avg=%9.1f us min=%9.1f max=%9.1f" % (r["Name"][:100], r["Calls"], float(r["AverageNs"])/1e3, float(r["MinNs"])/1e3, float(r["MaxNs"])/1e3))
PY
tail -2 gpurun_out/prof_ecc.log
rm -rf gpurun_out/prof_ecc
