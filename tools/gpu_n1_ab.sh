# N > 1 loop on one GPU, one-rank RCCL group, deferred schedule: the arrangements side by side in ONE call (alternating twice)
set -x

B="--force-chunked --defer-exchange --steps 10 --warmup 3 --no-cpu-baseline --no-reraycast"
for rep in 1 2; do
  python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-reraycast 2>/dev/null | tail -1 > gpurun_out/ab_plain_$rep.json
  UPSP_FORCE_COLLECTIVES=1 UPSP_EXCHANGE_SELF_RCCL=1 python bench.py $B 2>/dev/null | tail -1 > gpurun_out/ab_selfrccl_$rep.json
  UPSP_FORCE_COLLECTIVES=1 UPSP_EXCHANGE_SELF_RCCL=1 UPSP_XFER_PRIORITY=low python bench.py $B 2>/dev/null | tail -1 > gpurun_out/ab_selfrccl_low_$rep.json
  UPSP_FORCE_COLLECTIVES=1 UPSP_EXCHANGE_SELF_RCCL=1 UPSP_BENCH_DRAIN_FIRST=1 python bench.py $B 2>/dev/null | tail -1 > gpurun_out/ab_selfrccl_drainfirst_$rep.json
  UPSP_FORCE_COLLECTIVES=1 python bench.py $B 2>/dev/null | tail -1 > gpurun_out/ab_inplace_$rep.json
  UPSP_FORCE_COLLECTIVES=1 python bench.py --force-chunked --sync-exchange --steps 10 --warmup 3 --no-cpu-baseline --no-reraycast 2>/dev/null | tail -1 > gpurun_out/ab_inplace_sync_$rep.json
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/ab_*.json")):
    try:
        d = json.loads(open(f).read())
        k = d["kernels"]
        print("%-45s step %.3f ms  passB %.3f  passA %.3f  gather %.3f  self_check %s" % (f[11:-5], d["ms_per_step"], k["node_rows_kernel"]["ms_per_step"],
              k["scan_compact_kernel"]["ms_per_step"], k.get("gather_pixel_rows_kernel", {}).get("ms_per_step", 0), d.get("exchange_self_check")))
    except Exception as e:
        print(f, "failed", e)
PY
cat gpurun_out/passb_probe.txt
