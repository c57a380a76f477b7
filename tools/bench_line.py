import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
keys=sys.argv[2].split(",") if len(sys.argv)>2 else ("hot_scan_kernel","gather_tile_kernel")
print(sys.argv[1], round(d["ms_per_step"],3), {k:round(v,3) for k,v in d["breakdown_ms"].items()}, {k:round(v["avg_launch_ms"],4) for k,v in d["kernels"].items() if any(x in k for x in keys)})
