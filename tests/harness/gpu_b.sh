mkdir -p gpurun_out/r3a
PAGRAPH_TIMING=1 timeout 600 python bench.py --steps 5 --warmup 1 --no-cpu-baseline > gpurun_out/r3a/bench.out 2> gpurun_out/r3a/bench.err
tail -2 gpurun_out/r3a/bench.err
python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r3a/bench.out') if l.startswith('{')][-1])
c=d['config']
print(d['value'], d['ms_per_step'], {k:round(v,1) for k,v in c.items() if k.startswith('ms_')})
PY
timeout 900 python -m pytest tests/test_gpu_big.py tests/test_gpu_shards.py -x -q -m gpu > gpurun_out/r3a/b.log 2>&1; tail -3 gpurun_out/r3a/b.log
