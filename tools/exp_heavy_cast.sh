#!/bin/bash
# heavy hand-off threshold of the batch ray caster on the bench's pixel rays (tunnel / UV-sphere / frame-filling sphere)
for s in 0 128 160 192 256; do
  for m in "" "uv" "fill"; do
    UPSP_HEAVY_STEPS_CAST=$s timeout -k 10 200 python tools/exp_pixelrays.py $m 2>&1 | grep -v amdgpu | head -1 | sed "s/^/steps=$s /"
  done
done
