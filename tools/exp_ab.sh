#!/bin/bash
# A/B of the frame-loop kernels with the long (reference order) and the short projection build in front of them
ext='import json,sys
d=json.loads(sys.stdin.readlines()[-1]); k=d["kernels"]
print(sys.argv[1], "step %.3f build %.3f loop %.3f | scan %.3f rows %.3f" % (d["ms_per_step"], d["breakdown_ms"]["projection_build"], d["breakdown_ms"]["frame_loop"], k["scan_compact_kernel"]["ms_per_step"], k["node_rows_kernel"]["ms_per_step"]))'
for i in 1 2 3; do
  for c in 0 1; do
    UPSP_OBLIQUE_CULL=$c timeout -k 10 200 python bench.py --no-cpu-baseline --no-reraycast 2>/dev/null | python -c "$ext" "cull=$c"
  done
done
