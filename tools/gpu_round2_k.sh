python -m pytest tests/test_frames_gpu.py tests/test_configs_gpu.py tests/test_psp_gpu.py tests/test_cli.py -x -q -m gpu 2>&1 | tail -5
timeout -k 10 400 python tools/scale_5m.py 2>&1 | grep "frame loop\|kernels"
