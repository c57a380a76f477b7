# Round-6 A/B runs, one gpurun call per part (from the repository root on the GPU box).  Lines land in gpurun_out/r06/.
#   bash tools/r06_ab.sh tests     GPU test suite
#   bash tools/r06_ab.sh bins      length-binned primary lists on / off (UPSP_RAY_BINS), hand-off threshold 96 / 48, alternating
#   bash tools/r06_ab.sh stagger   start delay of the general ECC iteration's workgroups (UPSP_ECC_STAGGER), --registration
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
pick = {n: round(v["avg_launch_ms"], 4) for n, v in k.items() if any(s in n for s in ("primary", "retry>", "scan", "node_rows", "ecc_sums", "gauss", "primary_list"))}
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
elif [ $part = stagger ]; then
  for i in 1 2; do
    for s in 0 3 6 12; do
      line stagger_${s}_$i UPSP_ECC_STAGGER=$s -- --registration --steps 3 --warmup 1 || exit 1
    done
  done
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
if [ $part = multi ]; then
  # several-camera row pass (configs[4] shape, 4 cameras x 5 M triangles): UPSP_MULTI_VARIANT 0 = round 5, 1 = padded rows,
  # 2 = padded + all series loads first, 3 = all loads first
  for i in 1 2; do
    for v in 0 1 2 3; do
      timeout -k 10 600 env UPSP_MULTI_VARIANT=$v python3 bench.py --cameras 4 --model 5m --steps 3 --warmup 1 --no-cpu-baseline > $o/multi_${v}_$i.json 2>> $o/err.log || { tail -5 $o/err.log; exit 1; }
      python3 - $o/multi_${v}_$i.json $v <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
k = d["kernels"]
print("variant %s: step %.3f ms, %.0f frame sets/s, node_rows_multi %.4f ms, scan %.4f ms" % (sys.argv[2], d["ms_per_step"], d["value"],
      k["node_rows_multi_kernel"]["avg_launch_ms"], k.get("scan_compact_kernel", {}).get("ms_per_step", 0)), flush=True)
PY
    done
  done
fi
