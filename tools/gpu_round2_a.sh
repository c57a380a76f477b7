set -o pipefail
python -m pytest tests/test_frames_gpu.py tests/test_psp_gpu.py -x -q -m gpu > gpurun_out/t_frames.log 2>&1; echo "frames rc=$?" 
tail -5 gpurun_out/t_frames.log
for r in 4 8; do
 for ld in 1024; do
  UPSP_ROWS_PER_WG=$r UPSP_BENCH_LD=$ld timeout -k 10 120 python bench.py --no-cpu-baseline --no-reraycast --steps 10 --warmup 3 > gpurun_out/b_rows_${r}_${ld}.log 2>&1
  python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/b_rows_${r}_${ld}.log").read().strip().splitlines()[-1])
    k=d["kernels"]
    print("rows/wg $r ld $ld: step %.3f ms fps %.0f frame_loop %.3f proj %.3f" % (d["ms_per_step"], d["value"], d["breakdown_ms"]["frame_loop"], d["breakdown_ms"]["projection_build"]), {n:(round(v["avg_launch_ms"]*1e3,1), v["calls_per_step"]) for n,v in k.items()})
except Exception as e: print("fail", e)
PY
 done
done
