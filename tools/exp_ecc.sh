#!/bin/bash
# ecc_sums2_kernel variants (pixels per thread and trip, waves per SIMD) on bench.py --registration
ext='import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(sys.argv[1], d["value"], d["ms_per_step"], {k:round(v["ms_per_step"],3) for k,v in d["kernels"].items() if "ecc" in k})'
run() { env "$@" timeout -k 10 300 python bench.py --registration --no-cpu-baseline --no-reraycast 2>/dev/null | python -c "$ext" "$*"; }
for v in 2 4 5 6 0; do run UPSP_ECC_VARIANT=$v; done
for v in 0 1 2; do run UPSP_ECC_IVARIANT=$v; done
