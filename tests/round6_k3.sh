#!/bin/bash
# Round 6: K3's similarity masks in the two-subtraction form: the segment kernels against the sequential restatement, build parity, the step
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_build_parity.py tests/test_gpu_configs.py -x -q -m gpu 2>&1 | tail -3
for rep in 1 2; do
python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-file-to-file --no-live-traffic 2> gpurun_out/k3.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().split('\n')[-1]); c=d['config']
print('ms_per_step=%.1f' % d['ms_per_step'], 'build=%.1f' % c['ms_build_device'], 'extract=%.1f sort=%.1f cluster=%.2f edges=%.2f' % (c['ms_extract'], c['ms_sort'], c['ms_cluster'], c['ms_edges']), 'succ=%.1f' % c['ms_successor_stage_wall'], 'walks=%.1f' % c['ms_walks_wall'], c['path_checksum'])
"
done
python bench.py --reads 250000 --ref-len 62500000 --steps 3 --warmup 1 --no-cpu-baseline --no-file-to-file --no-live-traffic 2> gpurun_out/k3_40x.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().split('\n')[-1]); c=d['config']
print('40x of 62.5 Mb: ms_per_step=%.1f' % d['ms_per_step'], 'build=%.1f' % c['ms_build_device'], 'cluster=%.2f edges=%.2f' % (c['ms_cluster'], c['ms_edges']))
"
