python -m pytest tests/ -x -q -m gpu > gpurun_out/t10.log 2>&1; echo rc=$? >> gpurun_out/t10.log; tail -5 gpurun_out/t10.log
