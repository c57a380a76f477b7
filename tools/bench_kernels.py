import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(sys.argv[1], "step %.3f ms" % d["ms_per_step"], {k:round(v,3) for k,v in d["breakdown_ms"].items()})
for k,v in sorted(d["kernels"].items(), key=lambda kv:-kv[1]["ms_per_step"]):
    print("  %-32s calls %6.1f  ms/step %8.3f  avg %.4f" % (k, v["calls_per_step"], v["ms_per_step"], v["avg_launch_ms"]))
