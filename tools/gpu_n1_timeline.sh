set -x
python -m pytest tests/test_imageops_gpu.py tests/test_psp_gpu.py tests/test_patch_setup_gpu.py tests/test_cli.py tests/test_bench_gpu.py -x -q -m gpu -s > gpurun_out/t3.log 2>&1; echo rc=$? >> gpurun_out/t3.log
tail -5 gpurun_out/t3.log
# N > 1 loop on one GPU through one-rank RCCL: plain default, then forced chunked deferred / in-step, with the device time line
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-reraycast > gpurun_out/n1_plain.json 2> gpurun_out/n1_plain.err
UPSP_FORCE_COLLECTIVES=1 UPSP_TRACE_TIMELINE=1 python bench.py --force-chunked --defer-exchange --steps 10 --warmup 3 --no-cpu-baseline --no-reraycast > gpurun_out/n1_defer.json 2> gpurun_out/n1_defer.err
UPSP_FORCE_COLLECTIVES=1 UPSP_TRACE_TIMELINE=1 python bench.py --force-chunked --steps 10 --warmup 3 --no-cpu-baseline --no-reraycast > gpurun_out/n1_sync.json 2> gpurun_out/n1_sync.err
python - <<'PY'
import json
for n in ("n1_plain","n1_defer","n1_sync"):
    try:
        d=json.load(open("gpurun_out/%s.json"%n))
        print(n, d["ms_per_step"], {k:(round(v["ms_per_step"],4)) for k,v in d["kernels"].items()})
    except Exception as e: print(n, "failed", e)
PY
