#!/bin/bash
# Round 6: A/B of a bench switch: tests/round6_ab.sh VAR "v1 v2 ..." [bench args]
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
var=$1; vals=$2; shift 2
for v in $vals; do
  for rep in 1 2; do
    env $var=$v python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-file-to-file --no-live-traffic "$@" 2> gpurun_out/ab_$v.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().split('\n')[-1]); c=d['config']
print('$var=$v', 'ms_per_step=%.1f' % d['ms_per_step'], 'prepare_wait=%.1f' % c['ms_prepare_wall'], 'prepare_beside=%.1f' % c.get('ms_prepare_beside_the_previous_walks', 0), 'build=%.1f' % c['ms_build_device'], 'succ=%.1f' % c['ms_successor_stage_wall'], 'walks=%.1f' % c['ms_walks_wall'], 'wait_host=%.1f' % c['ms_wait_for_previous_host_half'], c['path_checksum'])
" | tee -a gpurun_out/r06_ab_$var.txt
  done
done
