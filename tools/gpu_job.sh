python -m pytest tests/test_cli.py -x -q -m gpu > gpurun_out/t17.log 2>&1; echo rc=$? >> gpurun_out/t17.log; tail -30 gpurun_out/t17.log
