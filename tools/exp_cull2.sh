#!/bin/bash
ext='import json,sys
d=json.loads(sys.stdin.readlines()[-1]); k=d["kernels"]
print(sys.argv[1], "build %.3f ms" % d["breakdown_ms"]["projection_build"], " ".join("%s %.3f" % (n.replace("projection_kernel",""), k[n]["ms_per_step"]) for n in k if "projection_kernel" in n or "heavy" in n or "witness" in n))'
run() { env "$@" timeout -k 10 200 python bench.py --no-cpu-baseline --no-reraycast 2>/dev/null | python -c "$ext" "$*"; }
for b in 1 2 3 4 6 8; do run UPSP_BLOCKS_PER_CU=$b; done
for b in 1 2 4; do run UPSP_BLOCKS_PER_CU=$b UPSP_REFILL=56; done
for b in 1 2 4; do run UPSP_BLOCKS_PER_CU=$b UPSP_REFILL=24; done
