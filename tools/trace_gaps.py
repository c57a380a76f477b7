"""Largest idle gaps in a rocprofv3 --kernel-trace csv (run: python3 tools/trace_gaps.py <dir> [n]): for each, the kernels either side.
Used to tell a host-side pause (nothing running) from one long kernel."""
import csv, glob, sys
d, n = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 8
rows = []
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:70], r.get("Queue_Id", "?")))
rows.sort()
print(len(rows), "kernel launches")
gaps, end, last = [], None, None
for s, e, k, q in rows:
    if end is not None and s > end:
        gaps.append((s - end, last, k, q))
    if end is None or e > end:
        end, last = e, k
for g in sorted(gaps, reverse=True)[:n]:
    print("idle %9.3f ms   after %-70s   before %s (queue %s)" % (g[0] / 1e6, g[1], g[2], g[3]))
longest = sorted(rows, key=lambda r: r[0] - r[1])[:n]
for s, e, k, q in longest:
    print("long %9.3f ms   %s (queue %s)" % ((e - s) / 1e6, k, q))
