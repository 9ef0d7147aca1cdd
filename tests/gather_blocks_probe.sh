#!/bin/bash
# GPU box: walks' wall time against the grid of the delivery kernel that runs beside them (PAG_GATHER_BLOCKS), no trace
out=${1:-gpurun_out/gather_blocks_probe.txt}
: > $out
for v in "PAG_GATHER_BLOCKS=4096" "PAG_GATHER_BLOCKS=16" "PAG_GATHER_BLOCKS=32" "PAG_GATHER_BLOCKS=8" "PAG_GATHER_BLOCKS=4096" "PAG_GATHER_BLOCKS=16" "PAG_GATHER_BLOCKS=16 PAG_WALK_PRIO=0"; do
  env $v python bench.py --steps 10 --warmup 1 --no-live-traffic --no-file-to-file --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
r = json.loads(sys.stdin.readlines()[-1]); c = r['config']
print('$v', 'ms_per_step', round(r['ms_per_step'], 1), 'walks', round(c['ms_walks_wall'], 1), 'successor stage', round(c['ms_successor_stage_wall'], 1), 'host epilogue', round(c['ms_traverse_host_epilogue'], 1), c['path_checksum'])" | tee -a $out
done
