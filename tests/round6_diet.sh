#!/bin/bash
# Round 6: the instruction / address-space diet (emit kernel's accept bookkeeping, LDS wave state as ds ops, the walker's job buffers as
# global loads, k_view_mark's searches, forward codes by one reversal): parity tests that cover them, then the step and the kernels
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_succ_golden.py tests/test_gpu_succ_modes.py tests/test_gpu_build_parity.py tests/test_gpu_kmer_counter.py tests/test_gpu_cli.py tests/test_gpu_configs.py -x -q -m gpu 2>&1 | tail -3
for rep in 1 2; do
python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-file-to-file --no-live-traffic 2> gpurun_out/diet.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().split('\n')[-1]); c=d['config']
print('ms_per_step=%.1f' % d['ms_per_step'], 'prepare=%.1f' % c['ms_prepare_wall'], 'build=%.1f' % c['ms_build_device'], 'extract=%.1f sort=%.1f cluster=%.2f edges=%.2f' % (c['ms_extract'], c['ms_sort'], c['ms_cluster'], c['ms_edges']), 'succ=%.1f' % c['ms_successor_stage_wall'], 'walks=%.1f' % c['ms_walks_wall'], c['path_checksum'])
"
done
ARGS="--reps 3" tests/succ_stage_probe.sh 2>&1 | grep -v rocprofv3 | head -16
