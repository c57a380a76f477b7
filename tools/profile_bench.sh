#!/bin/bash
# rocprofv3 passes for bench.py on the GPU box: kernel-trace stats + HBM traffic counters
# (FETCH_SIZE and WRITE_SIZE in separate --pmc passes, MI355X_MICROARCH.md "rocprofv3 PMC slots").
# usage: tools/profile_bench.sh <tag> [bench args...]
tag=$1; shift
out=gpurun_out/prof_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 240 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 bench.py --no-cpu-baseline --no-reraycast "$@" > $out/trace.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 240 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $out/$c -- python3 bench.py --no-cpu-baseline --no-reraycast "$@" > $out/$c.log 2>&1
done
python3 - $out <<'PY'
import csv, glob, sys, collections, json
import re
out = sys.argv[1]
res = collections.OrderedDict()
def kname(full):
    m = re.search(r"\(anonymous namespace\)::(\w+)(<[^>]*>)?\(", full)
    return (m.group(1) + (m.group(2) or "")) if m else full[:60]
f = glob.glob(out + "/trace/*/*kernel_stats.csv")
if f:
    for r in csv.DictReader(open(f[0])):
        if "upsp" in r["Name"]:
            name = kname(r["Name"])
            res.setdefault(name, {}).update(calls=int(r["Calls"]), avg_ns=float(r["AverageNs"]), total_ns=int(r["TotalDurationNs"]))
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob(out + "/%s/*/*counter_collection.csv" % c)
    if not f: continue
    acc = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(f[0])):
        if "upsp" in r["Kernel_Name"] and r["Counter_Name"] == c:
            name = kname(r["Kernel_Name"])
            acc[name][0] += 1; acc[name][1] += float(r["Counter_Value"])
    for name, (n, v) in acc.items():
        res.setdefault(name, {})[c + "_KB_per_launch"] = v / n
json.dump(res, open(out + "/summary.json", "w"), indent=1)
for k, v in res.items(): print(k, v)
PY
