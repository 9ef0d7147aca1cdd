mkdir -p gpurun_out/r3a
timeout 1200 python -m pytest tests/test_gpu_prepare.py tests/test_gpu_cli.py tests/test_gpu_build_parity.py tests/test_gpu_configs.py -x -q -m gpu > gpurun_out/r3a/prep.log 2>&1; tail -15 gpurun_out/r3a/prep.log
