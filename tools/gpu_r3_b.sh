set -o pipefail
mkdir -p gpurun_out/r3b
timeout -k 10 500 python3 bench.py > gpurun_out/r3b/bench_line.json 2> gpurun_out/r3b/bench_line.err; echo "bench rc=$?"; tail -3 gpurun_out/r3b/bench_line.err
timeout -k 10 300 python3 bench.py --registration > gpurun_out/r3b/bench_line_reg.json 2> gpurun_out/r3b/bench_line_reg.err; echo "bench reg rc=$?"; tail -3 gpurun_out/r3b/bench_line_reg.err
python -m pytest tests -x -q -m gpu > gpurun_out/r3b/gpu_tests.log 2>&1; echo "all tests rc=$?"; tail -4 gpurun_out/r3b/gpu_tests.log
