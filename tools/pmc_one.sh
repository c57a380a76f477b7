#!/bin/bash
# one PMC pass (counters only) over a python script; prints per-kernel averages
# usage: tools/pmc_one.sh <outdir> "<counter list>" <script.py> [args...]
out=$1; grp=$2; shift; shift
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout ${PMC_TIMEOUT:-150} rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $out -- python3 "$@" > $out.log 2>&1
f=$(find $out -name "*counter_collection.csv" | head -1)
[ -n "$f" ] && python3 - "$f" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.OrderedDict()
for r in rows:
    k = r["Kernel_Name"]
    if "upsp" not in k: continue
    import re
    m = re.search(r"\(anonymous namespace\)::(\w+)(<[^>]*>)?\(", k)
    short = (m.group(1) + (m.group(2) or "")) if m else k[:40]
    key = (short, r["Counter_Name"])
    a = agg.setdefault(key, [0, 0.0, 0.0])
    a[0] += 1; a[1] += float(r["Counter_Value"]); a[2] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
for (k, c), (n, v, t) in agg.items():
    print("%-38s %-32s calls=%d avg=%.6g avg_ns=%.0f" % (k[:38], c, n, v / n, t / n))
PY
