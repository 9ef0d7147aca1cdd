# development aid: the whole GPU suite, then a bench line
mkdir -p gpurun_out/r3a
timeout 2400 python -m pytest tests -x -q -m gpu > gpurun_out/r3a/full.log 2>&1; tail -12 gpurun_out/r3a/full.log | cut -c1-300
PAGRAPH_TIMING=1 timeout 600 python bench.py --steps 5 --warmup 1 --no-cpu-baseline > gpurun_out/r3a/bench.out 2> gpurun_out/r3a/bench.err
grep "pag_travel laps\|examined a vertex\|successor records" gpurun_out/r3a/bench.err | tail -4 | cut -c1-400
python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r3a/bench.out') if l.startswith('{')][-1])
c=d['config']
print(d['value'], d['ms_per_step'], {k:round(v,1) for k,v in c.items() if k.startswith('ms_')})
PY
