set -o pipefail
mkdir -p gpurun_out/r3g
python -m pytest tests/test_exchange_gpu.py tests/test_bench_gpu.py tests/test_cli.py tests/test_configs_gpu.py -x -q -m gpu > gpurun_out/r3g/tests.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -12 gpurun_out/r3g/tests.log
[ $rc -eq 0 ] || exit 1
UPSP_FORCE_COLLECTIVES=1 timeout -k 10 300 python3 bench.py --force-chunked --no-cpu-baseline --no-reraycast > gpurun_out/r3g/chunked_rccl.json 2> gpurun_out/r3g/chunked_rccl.err; echo "chunked rccl rc=$?"; tail -2 gpurun_out/r3g/chunked_rccl.err
UPSP_FORCE_COLLECTIVES=1 timeout -k 10 300 python3 bench.py --force-chunked --wire12 --no-cpu-baseline --no-reraycast > gpurun_out/r3g/chunked_rccl12.json 2> gpurun_out/r3g/chunked_rccl12.err; echo "chunked rccl 12 rc=$?"; tail -2 gpurun_out/r3g/chunked_rccl12.err
timeout -k 10 300 python3 bench.py --force-chunked > gpurun_out/r3g/chunked.json 2> gpurun_out/r3g/chunked.err; echo "chunked rc=$?"
python3 - <<'PY'
import json
for n in ("chunked_rccl","chunked_rccl12","chunked"):
    try:
        d=json.loads(open("gpurun_out/r3g/%s.json" % n).read().strip().splitlines()[-1])
        print(n, round(d["value"]), round(d["ms_per_step"],3), d["breakdown_ms"], d.get("exchange_bytes_per_step"), d.get("parity_checked"))
        print("   ", {k: round(v["ms_per_step"],3) for k,v in d["kernels"].items() if v["ms_per_step"]>0.02})
    except Exception as e: print(n, "failed", e)
PY
