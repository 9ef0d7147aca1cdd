#!/bin/bash
# GPU box: the last stretch of every strand in half-length segments (PAG_SEG_TAIL_FRAC) against the walks' wall time
out=${1:-gpurun_out/seg_tail_probe.txt}
: > $out
for v in "X=1" "PAG_SEG_TAIL_FRAC=0.15" "PAG_SEG_TAIL_FRAC=0.25" "PAG_SEG_TAIL_FRAC=0.4" "X=1" "PAG_SEG_TAIL_FRAC=0.25"; do
  env $v python bench.py --steps 10 --warmup 1 --no-live-traffic --no-file-to-file --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
r = json.loads(sys.stdin.readlines()[-1]); c = r['config']
print('$v', 'ms_per_step', round(r['ms_per_step'], 1), 'walks', round(c['ms_walks_wall'], 1), 'jobs', c['walk_jobs'], 'classifications', c['walk_classifications'], c['path_checksum'])" | tee -a $out
done
