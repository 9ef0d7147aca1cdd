mkdir -p gpurun_out/r3a
timeout 900 python -m pytest tests/test_gpu_shards.py -x -q -m gpu -k "two_processes or rccl" > gpurun_out/r3a/shx.log 2>&1; tail -30 gpurun_out/r3a/shx.log | cut -c1-250
