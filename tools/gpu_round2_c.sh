set -o pipefail
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/prof_chunked
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_chunked -- python3 bench.py --force-chunked --no-cpu-baseline --no-reraycast --steps 3 --warmup 1 > gpurun_out/prof_chunked.log 2>&1; echo "rc=$?"
f=$(find gpurun_out/prof_chunked -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: -int(r["TotalDurationNs"]))
for r in rows[:25]:
    print("%-90s calls=%5s avg=%9.1f us total=%9.1f ms" % (r["Name"][:90], r["Calls"], float(r["AverageNs"])/1e3, int(r["TotalDurationNs"])/1e6))
PY
tail -3 gpurun_out/prof_chunked.log | cut -c1-600
