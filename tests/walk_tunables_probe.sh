#!/bin/bash
# GPU box: the walks' tunables re-measured with the deliveries bounded (round 5)
out=${1:-gpurun_out/walk_tunables_probe.txt}
: > $out
for v in "X=1" "PAG_POST_INTERLEAVE=32" "PAG_POST_INTERLEAVE=64" "PAG_POST_INTERLEAVE=128" "PAG_WALK_WAVES_PER_CU=5" "PAG_WALK_WAVES_PER_CU=6" "PAG_STITCH_THREADS=4" "PAG_GATHER_BLOCKS=48" "PAG_POST_INTERLEAVE=32" "X=1"; do
  env $v python bench.py --steps 10 --warmup 1 --no-live-traffic --no-file-to-file --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
r = json.loads(sys.stdin.readlines()[-1]); c = r['config']
print('$v', 'ms_per_step', round(r['ms_per_step'], 1), 'walks', round(c['ms_walks_wall'], 1), 'successor stage', round(c['ms_successor_stage_wall'], 1), c['path_checksum'])" | tee -a $out
done
