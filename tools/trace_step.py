"""Timeline of one steady-state step from a rocprofv3 --kernel-trace csv: the launches between two consecutive launches of an
anchor kernel (default node_rows_kernel; the last two, or the pair that starts at occurrence `k`), start / duration in
microseconds and the queue.   python3 tools/trace_step.py <dir> [anchor] [k]"""
import csv, glob, sys
d = sys.argv[1]
anchor = sys.argv[2] if len(sys.argv) > 2 else "node_rows_kernel"
rows = []
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "?")))
rows.sort()
idx = [i for i, r in enumerate(rows) if anchor in r[2]]
k = int(sys.argv[3]) if len(sys.argv) > 3 else len(idx) - 2
a, b = idx[k], idx[k + 1]
t0 = rows[a][1]
for s, e, k, q in rows[a:b + 1]:
    k = k.replace("upsp::(anonymous namespace)::", "").replace("(anonymous namespace)::", "").replace("void ", "")
    print("%9.1f us  +%8.1f us  q%-3s %s" % ((s - t0) / 1e3, (e - s) / 1e3, q, k[:100]))
