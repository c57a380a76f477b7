#!/bin/bash
# Projection build after the early oblique test (10x fewer rays: the kernels are bound by their longest rays now):
# switches that did not pay while the traversal was throughput-bound, measured again.
ext='import json,sys
d=json.loads(sys.stdin.readlines()[-1]); k=d["kernels"]
print(sys.argv[1], "build %.3f ms" % d["breakdown_ms"]["projection_build"], " ".join("%s %.3f" % (n.replace("projection_kernel",""), k[n]["ms_per_step"]) for n in k if "projection_kernel" in n or "heavy" in n or "witness" in n))'
run() { env "$@" timeout -k 10 200 python bench.py --no-cpu-baseline --no-reraycast 2>/dev/null | python -c "$ext" "$*"; }
run A=0
run UPSP_PREFETCH=1
run UPSP_TOUCH=1
run UPSP_HEAVY_STEPS=96
run UPSP_HEAVY_STEPS=64
run UPSP_HEAVY_STEPS=48
run UPSP_REFILL=20
run UPSP_REFILL=56
run UPSP_DESC_CAP=2
run UPSP_DESC_CAP=12
run UPSP_XCD_AWARE=0
run UPSP_NO_WITNESS=1
run UPSP_OBLIQUE_CULL=0
# blocks per CU / refill thresholds (all rays resident at once wins)
for b in 1 2 3 4 6 8; do run UPSP_BLOCKS_PER_CU=$b; done
for b in 1 2 4; do run UPSP_BLOCKS_PER_CU=$b UPSP_REFILL=56; done
for b in 1 2 4; do run UPSP_BLOCKS_PER_CU=$b UPSP_REFILL=24; done
# own-triangle bound for the primary rays, residual descent cap
run A=0
run UPSP_OWN_BOUND_PRIMARY=1
run UPSP_OWN_BOUND_PRIMARY=1 UPSP_HEAVY_STEPS=64
run UPSP_NO_OWN_BOUND=1
run UPSP_DESC_CAP_RESIDUAL=2
run UPSP_DESC_CAP_RESIDUAL=8
