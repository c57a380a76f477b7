python -m pytest tests/test_configs_gpu.py tests/test_reference_pins_gpu.py tests/test_projection_gpu.py -x -q -m gpu --durations=8 > gpurun_out/t_cfg.log 2>&1; echo "rc=$?"
tail -25 gpurun_out/t_cfg.log
