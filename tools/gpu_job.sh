python -m pytest tests/test_exchange_gpu.py tests/test_bench_gpu.py -x -q -m gpu > gpurun_out/t16.log 2>&1; echo rc=$? >> gpurun_out/t16.log; tail -5 gpurun_out/t16.log
UPSP_FORCE_COLLECTIVES=1 UPSP_EXCHANGE_SELF_RCCL=1 python bench.py --force-chunked --defer-exchange --steps 10 --warmup 3 --no-cpu-baseline --no-reraycast 2>/dev/null | tail -1 > gpurun_out/b16.json
python - <<'PY'
import json
d=json.loads(open("gpurun_out/b16.json").read())
print(d["ms_per_step"], d["rccl_nranks"], d["rccl_bound"], d["exchange_self_check"])
PY
