#!/bin/bash
# Projection build: waves a ray list is spread over (UPSP_WAVES_PER_SIMD) and the heavy hand-off, bench model
ext='import json,sys
d=json.loads(sys.stdin.readlines()[-1]); k=d["kernels"]
print(sys.argv[1], "build %.3f ms" % d["breakdown_ms"]["projection_build"], " ".join("%s %.3f" % (n.replace("projection_kernel",""), k[n]["ms_per_step"]) for n in k if "projection_kernel" in n or "heavy" in n or "witness" in n or "list" in n))'
run() { env "$@" timeout -k 10 200 python bench.py --no-cpu-baseline --no-reraycast $MODEL 2>/dev/null | python -c "$ext" "$MODEL $*"; }
for w in 0 1 2 3 4 6; do run UPSP_WAVES_PER_SIMD=$w; done
run UPSP_HEAVY_STEPS=64
run UPSP_DESC_CAP_RESIDUAL=8
run UPSP_DESC_CAP_RESIDUAL=1
MODEL="--model uv"
for w in 0 2 3 4; do run UPSP_WAVES_PER_SIMD=$w; done
