#!/bin/bash
# Finer sweep of UPSP_HEAVY_STEPS with the number of rays handed over (see exp_heavy.sh)
ext='import json,sys
d=json.loads(sys.stdin.readlines()[-1]); k=d["kernels"]
print(sys.argv[1], "build %.3f ms" % d["breakdown_ms"]["projection_build"], " ".join("%s %.3f" % (n.replace("projection_kernel",""), k[n]["ms_per_step"]) for n in k if "projection_kernel" in n or "heavy" in n or "witness" in n))'
for model in uv quad; do
  for steps in 96 128 160 192 224 256 320; do
    UPSP_HEAVY_STEPS=$steps timeout -k 10 200 python bench.py --model $model --no-cpu-baseline 2>/dev/null | python -c "$ext" "$model steps=$steps"
    UPSP_HEAVY_STEPS=$steps UPSP_DEBUG_COUNTS=1 timeout -k 10 100 python tools/prof_proj.py $model 2>&1 | grep "heavy rays" | tail -1
  done
done
