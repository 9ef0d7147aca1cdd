#!/bin/bash
# GPU box: order in which the first rounds' segment jobs enter the ring, re-measured with the deliveries bounded (round 5)
out=${1:-gpurun_out/post_order_probe.txt}
: > $out
for v in "X=1" "PAG_POST_INTERLEAVE=0" "PAG_POST_SPREAD=0.5" "PAG_POST_SPREAD=0.25" "PAG_POST_PROPORTIONAL=0" "PAG_POST_SPREAD=0.1" "X=1"; do
  env $v python bench.py --steps 10 --warmup 1 --no-live-traffic --no-file-to-file --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
r = json.loads(sys.stdin.readlines()[-1]); c = r['config']
print('$v', 'ms_per_step', round(r['ms_per_step'], 1), 'walks', round(c['ms_walks_wall'], 1), 'successor stage', round(c['ms_successor_stage_wall'], 1), 'host epilogue', round(c['ms_traverse_host_epilogue'], 1), c['path_checksum'])" | tee -a $out
done
