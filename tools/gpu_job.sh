python -m pytest tests/test_cli.py -x -q -m gpu > gpurun_out/t12.log 2>&1; echo rc=$? >> gpurun_out/t12.log; tail -30 gpurun_out/t12.log
