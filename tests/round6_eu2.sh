#!/bin/bash
# Round 6: the walker built for two waves per SIMD (make WALK_EU=2: 12 registers spilled now that the job's buffers are scalar global pointers; ~35 in round 5): waves per CU
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
for w in default 4 5; do
  for rep in 1 2; do
    if [ $w = default ]; then unset PAG_WALK_WAVES_PER_CU; else export PAG_WALK_WAVES_PER_CU=$w; fi
    python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-file-to-file --no-live-traffic 2> gpurun_out/eu2.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().split('\n')[-1]); c=d['config']
print('waves_per_cu=$w', 'ms_per_step=%.1f' % d['ms_per_step'], 'build=%.1f' % c['ms_build_device'], 'succ=%.1f' % c['ms_successor_stage_wall'], 'walks=%.1f' % c['ms_walks_wall'], c['path_checksum'])
" | tee -a gpurun_out/r06_eu2_probe.txt
  done
done
