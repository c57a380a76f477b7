#!/bin/bash
ext='import json,sys
d=json.loads(sys.stdin.readlines()[-1]); k=d["kernels"]
print(sys.argv[1], "build %.3f ms" % d["breakdown_ms"]["projection_build"], " ".join("%s %.3f" % (n.replace("projection_kernel",""), k[n]["ms_per_step"]) for n in k if "projection_kernel" in n or "heavy" in n or "witness" in n))'
run() { env "$@" timeout -k 10 200 python bench.py --no-cpu-baseline --no-reraycast 2>/dev/null | python -c "$ext" "$*"; }
run A=0
run UPSP_OWN_BOUND_PRIMARY=1
run UPSP_OWN_BOUND_PRIMARY=1 UPSP_HEAVY_STEPS=64
run UPSP_NO_OWN_BOUND=1
run UPSP_DESC_CAP_RESIDUAL=2
run UPSP_DESC_CAP_RESIDUAL=8
