mkdir -p gpurun_out/r3a
timeout 900 python -m pytest tests/test_gpu_prepare.py -x -q -m gpu -k generator > gpurun_out/r3a/prep2.log 2>&1; tail -15 gpurun_out/r3a/prep2.log
PAGRAPH_TIMING=1 timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r3a/bench.out 2> gpurun_out/r3a/bench.err
tail -3 gpurun_out/r3a/bench.err
python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r3a/bench.out') if l.startswith('{')][-1])
c=d['config']
print(d['value'], d['ms_per_step'], {k:round(v,1) for k,v in c.items() if k.startswith('ms_')})
print(d['roofline'])
PY
bash tests/harness/gpu_prof.sh
