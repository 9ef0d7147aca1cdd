# development aid: the walker-grid tests, the two-rank shard test five times, and a bench line
mkdir -p gpurun_out/r3a
timeout 900 python -m pytest tests/test_gpu_cli.py -x -q -m gpu -k "grid or share" > gpurun_out/r3a/grid.log 2>&1; tail -5 gpurun_out/r3a/grid.log
for i in 1 2 3 4 5; do
  timeout 600 python -m pytest tests/test_gpu_shards.py -x -q -m gpu -k two_processes > gpurun_out/r3a/shard2_$i.log 2>&1; tail -1 gpurun_out/r3a/shard2_$i.log
done
PAGRAPH_TIMING=1 timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r3a/bench.out 2> gpurun_out/r3a/bench.err; tail -1 gpurun_out/r3a/bench.out | cut -c1-400
grep "pag_travel laps" gpurun_out/r3a/bench.err | tail -1
