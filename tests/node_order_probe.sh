#!/bin/bash
# GPU box: the traversal graph's nodes numbered by place (default) against by code (PAG_NODE_ORDER=code) at BASELINE configs[1]
out=${1:-gpurun_out/node_order_probe.txt}
: > $out
for m in code place code place; do
  PAG_NODE_ORDER=$m python bench.py --steps 8 --warmup 1 --no-live-traffic --no-file-to-file --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
r = json.loads(sys.stdin.readlines()[-1]); c = r['config']
print('PAG_NODE_ORDER=$m', 'ms_per_step', round(r['ms_per_step'], 1), 'successor stage', round(c['ms_successor_stage_wall'], 1), 'walks', round(c['ms_walks_wall'], 1), 'build', round(c['ms_build_device'], 1), c['path_checksum'])" | tee -a $out
done
