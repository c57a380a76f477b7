set -o pipefail
python -m pytest tests -x -q -m gpu > gpurun_out/gpu_tests.log 2>&1; echo "tests rc=$?"
tail -5 gpurun_out/gpu_tests.log
bash tools/profile_bench.sh r02a 2>&1 | tail -40
