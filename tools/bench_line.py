import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(sys.argv[1], round(d["ms_per_step"],3), {k:round(v,3) for k,v in d["breakdown_ms"].items()}, {k:round(v["avg_launch_ms"],4) for k,v in d["kernels"].items() if k in ("hot_scan_kernel","gather_tile_kernel","hot_fix_kernel")})
