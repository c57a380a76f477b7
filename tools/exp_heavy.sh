#!/bin/bash
# Sweep of the heavy-ray hand-off (UPSP_HEAVY_STEPS / UPSP_HEAVY_GRID) on the two bench models.
# usage (GPU box, repository root): bash tools/exp_heavy.sh > gpurun_out/heavy.log
ext='import json,sys
d=json.loads(sys.stdin.readlines()[-1]); k=d["kernels"]
print(sys.argv[1], "build %.3f ms" % d["breakdown_ms"]["projection_build"], " ".join("%s %.3f" % (n.replace("projection_kernel",""), k[n]["ms_per_step"]) for n in k if "projection_kernel" in n or "heavy" in n or "witness" in n))'
for model in uv quad; do
  for steps in 0 64 128 256 512; do
    UPSP_HEAVY_STEPS=$steps timeout -k 10 200 python bench.py --model $model 2>/dev/null | python -c "$ext" "$model steps=$steps"
  done
  for grid in 128 1024; do
    UPSP_HEAVY_GRID=$grid timeout -k 10 200 python bench.py --model $model 2>/dev/null | python -c "$ext" "$model grid=$grid"
  done
done
UPSP_DEBUG_COUNTS=1 timeout -k 10 100 python tools/prof_proj.py uv 2>&1 | tail -2
