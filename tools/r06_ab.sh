# Round-6 A/B runs, one gpurun call per part (from the repository root on the GPU box).  Lines land in gpurun_out/r06/.
#   bash tools/r06_ab.sh tests     GPU test suite
#   bash tools/r06_ab.sh bins      length-binned primary lists on / off (UPSP_RAY_BINS), hand-off threshold 96 / 48, alternating
#   bash tools/r06_ab.sh step      the one-call step: pass A on a stream of its own / slab filter / ray bins, one switch at a time
#   bash tools/r06_ab.sh rays      tools/r06_rays.py (1 Mi pixel rays + projection build alone, slab filter on / off) with ray bins off / on
#   bash tools/r06_ab.sh fused     ECC pre-blur fused with the identity iteration (default) / two kernels (UPSP_ECC_FUSED_BLUR=0)
#   (the ECC start-delay, ECC unroll / occupancy and several-camera row-pass experiments of this round were run with switches that
#    no longer exist: their numbers are in LAB_NOTES.md section 13)
set -o pipefail
part=${1:-tests}
o=gpurun_out/r06; mkdir -p $o
line() { # name, env..., -- bench args
  name=$1; shift
  envs=""; while [ "$1" != "--" ]; do envs="$envs $1"; shift; done; shift
  timeout -k 10 500 env $envs python3 bench.py --no-cpu-baseline "$@" > $o/$name.json 2>> $o/err.log || { echo "$name failed"; tail -5 $o/err.log; return 1; }
  python3 - $o/$name.json $name <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
k = d.get("kernels", {})
pick = {n: round(v["avg_launch_ms"], 4) for n, v in k.items() if any(s in n for s in ("primary", "retry>", "scan", "node_rows", "ecc_sums", "ecc_blur", "gauss", "primary_list"))}
print("%-22s %9.0f %s  step %.4f ms  %s  %s" % (sys.argv[2], d["value"], d["unit"], d["ms_per_step"], {a: round(b, 3) for a, b in d.get("breakdown_ms", {}).items()}, pick), flush=True)
pr = d.get("pixel_rays_fill")
if pr: print("    pixel_rays_fill %.4f ms  %.0f Mrays/s" % (pr["ms"], pr["mrays_per_s"]), flush=True)
PY
}
if [ $part = tests ]; then
  timeout -k 10 1000 python3 -m pytest tests -m gpu -x -q 2>&1 | tail -15
elif [ $part = bins ]; then
  for i in 1 2 3; do
    line bins_off_$i UPSP_RAY_BINS=0 -- --no-reraycast --steps 10 --warmup 3 || exit 1
    line bins_on_$i UPSP_RAY_BINS=1 -- --no-reraycast --steps 10 --warmup 3 || exit 1
    line bins_on_h48_$i UPSP_RAY_BINS=1 UPSP_HEAVY_STEPS=48 -- --no-reraycast --steps 10 --warmup 3 || exit 1
    line bins_off_h48_$i UPSP_RAY_BINS=0 UPSP_HEAVY_STEPS=48 -- --no-reraycast --steps 10 --warmup 3 || exit 1
  done
  line serial_off UPSP_RAY_BINS=0 -- --no-reraycast --serial --steps 10 --warmup 3 || exit 1
  line serial_on UPSP_RAY_BINS=1 -- --no-reraycast --serial --steps 10 --warmup 3 || exit 1
  line serial_on_h48 UPSP_RAY_BINS=1 UPSP_HEAVY_STEPS=48 -- --no-reraycast --serial --steps 10 --warmup 3 || exit 1
  line serial_on_h64 UPSP_RAY_BINS=1 UPSP_HEAVY_STEPS=64 -- --no-reraycast --serial --steps 10 --warmup 3 || exit 1
fi
if [ $part = step ]; then
  # the one-call step: pass A on its own stream (UPSP_STEP_SCAN_STREAM), slab filter, ray bins -- one switch at a time against all off
  B="UPSP_STEP_SCAN_STREAM=0 UPSP_RAY_BINS=0 UPSP_SLAB_FILTER=0"
  for i in 1 2 3; do
    line base_$i $B -- --no-reraycast --steps 20 --warmup 5 || exit 1
    line scanstream_$i UPSP_STEP_SCAN_STREAM=1 UPSP_RAY_BINS=0 UPSP_SLAB_FILTER=0 -- --no-reraycast --steps 20 --warmup 5 || exit 1
    line slab_$i UPSP_STEP_SCAN_STREAM=0 UPSP_RAY_BINS=0 UPSP_SLAB_FILTER=1 -- --no-reraycast --steps 20 --warmup 5 || exit 1
    line bins_$i UPSP_STEP_SCAN_STREAM=0 UPSP_RAY_BINS=1 UPSP_SLAB_FILTER=0 -- --no-reraycast --steps 20 --warmup 5 || exit 1
  done
elif [ $part = rays ]; then
  UPSP_RAY_BINS=0 python3 tools/r06_rays.py || exit 1
  UPSP_RAY_BINS=1 python3 tools/r06_rays.py || exit 1
fi
if [ $part = sides ]; then
  # the one-call step with one side stream (UPSP_STEP_ONE_SIDE=1) against two builds in flight (default)
  for i in 1 2 3; do
    line one_side_$i UPSP_STEP_ONE_SIDE=1 -- --no-reraycast --steps 20 --warmup 5 || exit 1
    line two_sides_$i -- --no-reraycast --steps 20 --warmup 5 || exit 1
  done
fi
if [ $part = adapt ]; then
  # hand-off launches only when the last build of the same view needed them (UPSP_HEAVY_ADAPT), and the two memsets folded into kernels
  for i in 1 2 3; do
    line adapt_off_$i UPSP_HEAVY_ADAPT=0 -- --no-reraycast --steps 20 --warmup 5 || exit 1
    line adapt_on_$i -- --no-reraycast --steps 20 --warmup 5 || exit 1
  done
fi
if [ $part = plain ]; then
  # default step, three runs (for before / after comparisons of a library change across two calls on the SAME box: not possible --
  # use with a switch, or read the kernel times)
  for i in 1 2 3; do line plain_$i -- --no-reraycast --steps 20 --warmup 5 || exit 1; done
fi
if [ $part = oneflush ]; then
  # ECC interior blocks with ONE float segment per row piece (double totals behind the loop; general iteration at four waves per
  # SIMD on a 35-row tile, identity at five) -- the default -- against the 32-row segments of rounds 3-5 (UPSP_ECC_ONE_FLUSH=0)
  for i in 1 2; do
    line oneflush_0_$i UPSP_ECC_ONE_FLUSH=0 -- --registration --steps 3 --warmup 1 || exit 1
    line oneflush_1_$i -- --registration --steps 3 --warmup 1 || exit 1
  done
fi
if [ $part = fused ]; then
  # the ECC's pre-blur fused with the identity iteration's sums (ecc_blur_ident_kernel, the default) against the two kernels
  # (UPSP_ECC_FUSED_BLUR=0), alternating
  for i in 1 2; do
    line fused_0_$i UPSP_ECC_FUSED_BLUR=0 -- --registration --steps 3 --warmup 1 || exit 1
    line fused_1_$i -- --registration --steps 3 --warmup 1 || exit 1
  done
fi
if [ $part = pairs ]; then
  # general ECC iteration: taps and footprint arithmetic of two rows as packed pairs (default) / row by row (UPSP_ECC_PAIRS=0)
  for i in 1 2 3; do
    line pairs_0_$i UPSP_ECC_PAIRS=0 -- --registration --steps 3 --warmup 1 || exit 1
    line pairs_1_$i -- --registration --steps 3 --warmup 1 || exit 1
  done
fi
if [ $part = fusedw ]; then
  for i in 1 2; do
    line fusedw_2k_$i UPSP_ECC_FUSED_BLUR=0 -- --registration --steps 3 --warmup 1 || exit 1
    line fusedw_5_$i -- --registration --steps 3 --warmup 1 || exit 1
    line fusedw_4_$i UPSP_FUSED_W4=1 -- --registration --steps 3 --warmup 1 || exit 1
  done
fi
if [ $part = again ]; then
  # the second pass of the fused pre-blur on its own stream (default) / in line (UPSP_ECC_AGAIN_STREAM=0), configs[2]-sized runs
  for i in 1 2 3; do
    line again_0_$i UPSP_ECC_AGAIN_STREAM=0 -- --registration --frames 4096 --steps 3 --warmup 1 || exit 1
    line again_1_$i -- --registration --frames 4096 --steps 3 --warmup 1 || exit 1
  done
fi
