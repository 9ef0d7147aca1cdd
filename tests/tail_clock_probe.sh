#!/bin/bash
# GPU box: what slows the last walk jobs of a block (round 5): the deliveries of finished contigs beside them.  Walker wave priority
# (PAG_WALK_PRIO), the delivery kernel's grid (PAG_GATHER_BLOCKS), no deliveries beside the walks (PAG_DELIVER_EARLY=0);
# per-classification cost of the first-round segment jobs by the time they began
mkdir -p gpurun_out
for v in "PAG_WALK_PRIO=0" "X=1" "PAG_GATHER_BLOCKS=64" "PAG_GATHER_BLOCKS=16" "PAG_GATHER_BLOCKS=4" "PAG_WALK_PRIO=0 PAG_GATHER_BLOCKS=16" "PAG_DELIVER_EARLY=0"; do
  echo "== $v"
  env $v PAG_WALK_TRACE=1 python bench.py --steps 8 --warmup 1 --no-cpu-baseline --no-file-to-file --no-live-traffic 2> gpurun_out/tail_probe.log | python -c "
import sys, json
r = json.loads(sys.stdin.readlines()[-1]); c = r['config']
print('ms_per_step', round(r['ms_per_step'], 1), 'walks', round(c['ms_walks_wall'], 1), 'successor stage', round(c['ms_successor_stage_wall'], 1), c['path_checksum'])"
  python tests/walk_trace.py gpurun_out/tail_probe.log 1024 2>/dev/null | cut -c1-200 | head -5
  python tests/walk_rate.py gpurun_out/tail_probe.log
done
