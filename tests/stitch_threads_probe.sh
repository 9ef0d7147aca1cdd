#!/bin/bash
# GPU box: walks' wall time against the host threads that make the adoptions (PAG_STITCH_THREADS) at BASELINE configs[1]
out=${1:-gpurun_out/stitch_threads_probe.txt}
: > $out
for n in 1 8 1 8 4 16; do
  PAG_STITCH_THREADS=$n python bench.py --steps 10 --warmup 1 --no-live-traffic --no-file-to-file --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
r = json.loads(sys.stdin.readlines()[-1]); c = r['config']
print('PAG_STITCH_THREADS=$n', 'ms_per_step', round(r['ms_per_step'], 1), 'walks', round(c['ms_walks_wall'], 1), 'successor stage', round(c['ms_successor_stage_wall'], 1), c['path_checksum'])" | tee -a $out
done
