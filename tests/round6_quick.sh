#!/bin/bash
# Round 6: a quick look at the default bench (laps on), twice
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
for rep in 1 2; do
PAGRAPH_TIMING=1 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-file-to-file --no-live-traffic 2> gpurun_out/quick.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().split('\n')[-1]); c=d['config']
print('ms_per_step=%.1f' % d['ms_per_step'], 'prepare=%.1f' % c['ms_prepare_wall'], 'build=%.1f' % c['ms_build_device'], 'succ=%.1f' % c['ms_successor_stage_wall'], 'walks=%.1f' % c['ms_walks_wall'], 'wait_host=%.1f' % c['ms_wait_for_previous_host_half'], c['path_checksum'])
"
grep "pag_travel laps" gpurun_out/quick.err | tail -2 | cut -c1-420
done
