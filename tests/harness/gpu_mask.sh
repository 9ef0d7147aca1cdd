mkdir -p gpurun_out/r3p
O=$GRAFT_REPO_ROOT/gpurun_out/r3p
timeout 1200 python -m pytest tests/test_gpu_build_parity.py tests/test_gpu_big.py tests/test_gpu_configs.py tests/test_gpu_prepare.py -x -q -m gpu > $O/mask.log 2>&1; tail -3 $O/mask.log
python bench.py --file-to-file > $O/r03_bench_c2_line.json 2> $O/bench.err; python - <<PY
import json
d=json.load(open('$O/r03_bench_c2_line.json'))
print(d['value'], d['ms_per_step'], {k:round(v,1) for k,v in d['config'].items() if k.startswith('ms_')}, d['config'].get('file_to_file_live'))
PY
cd /tmp && export TMPDIR=/tmp
timeout 1500 python $GRAFT_REPO_ROOT/tests/k_sweep.py $O/r03_k_sweep_k16.json --ks 16 > $O/ksweep16.log 2>&1; tail -2 $O/ksweep16.log | cut -c1-500
