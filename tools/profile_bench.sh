#!/bin/bash
# rocprofv3 passes for bench.py on the GPU box: kernel-trace stats + HBM traffic counters
# (FETCH_SIZE and WRITE_SIZE in separate --pmc passes, MI355X_MICROARCH.md "rocprofv3 PMC slots").
# usage: tools/profile_bench.sh <tag> [bench args...]      (run from the repo root on the GPU box)
# writes gpurun_out/prof_<tag>/{kernel_stats.csv, summary.json} and, with the summary handed to
# bench.py (UPSP_BENCH_TRAFFIC_JSON), gpurun_out/prof_<tag>/bench_line.json
tag=$1; shift
out=gpurun_out/prof_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout -k 10 ${PROFILE_TIMEOUT:-300} rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 bench.py --no-cpu-baseline --no-reraycast "$@" > $out/trace.log 2>&1 || exit 1
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 ${PROFILE_TIMEOUT:-300} rocprofv3 --kernel-trace --pmc $c --output-format csv -d $out/$c -- python3 bench.py --no-cpu-baseline --no-reraycast "$@" > $out/$c.log 2>&1 || exit 1
done
python3 - $out "$@" <<'PY'
import csv, glob, sys, collections, json, re, shutil
out, args = sys.argv[1], sys.argv[2:]
res = collections.OrderedDict()
def kname(full):
    m = re.search(r"\(anonymous namespace\)::(\w+)(<[^>]*>)?\(", full)
    return (m.group(1) + (m.group(2) or "")) if m else full[:60]
f = glob.glob(out + "/trace/*/*kernel_stats.csv")
if f:
    shutil.copyfile(f[0], out + "/kernel_stats.csv")
    for r in csv.DictReader(open(f[0])):
        if "upsp" in r["Name"]:
            res.setdefault(kname(r["Name"]), {}).update(calls=int(r["Calls"]), avg_ns=float(r["AverageNs"]),
                                                         total_ns=int(r["TotalDurationNs"]))
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob(out + "/%s/*/*counter_collection.csv" % c)
    if not f: continue
    acc = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(f[0])):
        if "upsp" in r["Kernel_Name"] and r["Counter_Name"] == c:
            a = acc[kname(r["Kernel_Name"])]; a[0] += 1; a[1] += float(r["Counter_Value"])
    for name, (n, v) in acc.items():
        res.setdefault(name, {})[c + "_KB_per_launch"] = v / n
# streaming = wide coalesced reads, for which FETCH_SIZE reports half the bytes on gfx950 (MI355X_MICROARCH.md "HBM");
# pointer-chasing reads of the traversal kernels are taken as reported
def bench_key(name):
    """kernel (as rocprof names it) -> (name in bench.py's `kernels`, streaming?)"""
    for prefix, key, stream in (("scan_compact_kernel", "scan_compact_kernel", True), ("hot_scan_kernel", "hot_scan_kernel", True),
                                ("projection_kernel<false, 0", "projection_kernel<primary>", False),
                                ("projection_kernel<false, 2", "projection_kernel<retry>", False), ("witness_kernel", "witness_kernels", False),
                                ("node_rows_multi_kernel", "node_rows_multi_kernel", True), ("node_rows_kernel", "node_rows_kernel", True),
                                ("ecc_cols_kernel", "ecc_sums_kernel", True), ("ecc_blur_ident_kernel", "ecc_blur_ident_kernel", True),
                                ("gauss5_quad_kernel", "gauss_pass_kernels", True),
                                ("gauss_fused_kernel<unsigned short", "gauss_pass_kernels", True), ("warp_compact_kernel", "warp_u16_kernel", False),
                                ("warp_u16_kernel", "warp_u16_kernel", True), ("gather_tile", "gather_tile_kernel", True)):
        if name.startswith(prefix): return key, stream
    return None, False
traffic, weight = {}, {}
for name, v in res.items():
    key, stream = bench_key(name)
    keys = [key] if key else []
    # the two ECC sums kernels also under their own bench names (the roofline names the dominant one, not the blend)
    if name.startswith("ecc_cols_kernel<true"): keys.append("ecc_sums_identity")
    if name.startswith("ecc_cols_kernel<false"): keys.append("ecc_sums_general")
    for key in keys:
        if "FETCH_SIZE_KB_per_launch" in v and "WRITE_SIZE_KB_per_launch" in v:
            # several kernels under one bench name (the identity and the general ECC sums launches): call-weighted mean per launch
            b = ((2 if stream else 1) * v["FETCH_SIZE_KB_per_launch"] + v["WRITE_SIZE_KB_per_launch"]) * 1024
            n = v.get("calls", 1)
            traffic[key] = (traffic.get(key, 0.0) * weight.get(key, 0) + b * n) / (weight.get(key, 0) + n)
            weight[key] = weight.get(key, 0) + n
# pass A in two launches (upsp_pipeline_set_scan_split): one "launch" of bench.py's scan_compact_kernel entry = the sum of the two
parts = [res[k] for k in res if k.startswith(("scan_inactive_kernel", "scan_active_kernel"))]
if parts and all("FETCH_SIZE_KB_per_launch" in v and "WRITE_SIZE_KB_per_launch" in v for v in parts):
    traffic["scan_compact_kernel"] = sum((2 * v["FETCH_SIZE_KB_per_launch"] + v["WRITE_SIZE_KB_per_launch"]) * 1024 for v in parts)
summary = {"bench_args": " ".join(args), "kernels": res, "traffic_bytes_per_launch": traffic,
           "note": "rocprofv3 --kernel-trace --stats (durations) and two separate --pmc passes (FETCH_SIZE, WRITE_SIZE, unit KB); "
                   "traffic = FETCH x 2 for the streaming kernels (gfx950 reports half of wide coalesced reads) + WRITE"}
json.dump(summary, open(out + "/summary.json", "w"), indent=1)
for d in ("trace", "FETCH_SIZE", "WRITE_SIZE"):      # raw traces are large: keep the summaries only
    shutil.rmtree(out + "/" + d, ignore_errors=True)
for k, v in res.items(): print(k, v)
PY
UPSP_BENCH_TRAFFIC_JSON=$out/summary.json timeout -k 10 ${BENCH_TIMEOUT:-500} python3 bench.py "$@" > $out/bench_line.json 2> $out/bench_line.err || exit 1
python3 - $out <<'PY'
import json, sys
d = json.loads(open(sys.argv[1] + "/bench_line.json").read().strip().splitlines()[-1])
print({k: d.get(k) for k in ("value", "ms_per_step", "mrays_per_s", "breakdown_ms", "summary")}); print(d["roofline"]); print(d.get("cpu_baseline")); print(d.get("parity"))
PY
