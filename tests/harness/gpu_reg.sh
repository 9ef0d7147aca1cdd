mkdir -p gpurun_out/r3a
timeout 1400 python -m pytest tests/test_gpu_shards.py -x -q -m gpu -s > gpurun_out/r3a/reg.log 2>&1; tail -25 gpurun_out/r3a/reg.log | cut -c1-300
