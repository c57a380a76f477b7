# round 3, first GPU call: new tests first, then the whole GPU suite, smoke, the default bench line and configs[2]
set -o pipefail
mkdir -p gpurun_out/r3a
python -m pytest tests/test_imageops_gpu.py tests/test_raycast_gpu.py tests/test_frames_gpu.py tests/test_video.py -x -q -m gpu -s > gpurun_out/r3a/new_tests.log 2>&1; rc=$?; echo "new tests rc=$rc"; tail -5 gpurun_out/r3a/new_tests.log
[ $rc -eq 0 ] || exit 1
timeout -k 10 500 python3 bench.py > gpurun_out/r3a/bench_line.json 2> gpurun_out/r3a/bench_line.err; echo "bench rc=$?"; tail -3 gpurun_out/r3a/bench_line.err
timeout -k 10 300 python3 bench.py --registration > gpurun_out/r3a/bench_line_reg.json 2> gpurun_out/r3a/bench_line_reg.err; echo "bench reg rc=$?"; tail -3 gpurun_out/r3a/bench_line_reg.err
python -m pytest tests -x -q -m gpu > gpurun_out/r3a/gpu_tests.log 2>&1; echo "all tests rc=$?"; tail -4 gpurun_out/r3a/gpu_tests.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
