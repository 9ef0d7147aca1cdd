#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_predicates.py tests/test_gpu_succ_golden.py tests/test_gpu_succ_modes.py -x -q -m gpu 2>&1 | tail -3
ARGS="--reps 3" tests/succ_stage_probe.sh 2>&1 | grep -v rocprofv3 | head -5
python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-file-to-file --no-live-traffic 2> gpurun_out/diet.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().split('\n')[-1]); c=d['config']
print('ms_per_step=%.1f' % d['ms_per_step'], 'build=%.1f' % c['ms_build_device'], 'succ=%.1f' % c['ms_successor_stage_wall'], 'walks=%.1f' % c['ms_walks_wall'], c['path_checksum'])
"
