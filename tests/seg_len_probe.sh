#!/bin/bash
# GPU box: segment length of the first rounds' jobs against the walks' wall time, re-measured with the deliveries bounded (round 5)
out=${1:-gpurun_out/seg_len_probe.txt}
: > $out
for v in "X=1" "PAG_SEG_LEN=9000" "PAG_SEG_LEN=8000" "PAG_SEG_LEN=6000" "PAG_SEG_LEN=8000 PAG_SEG_OVERLAP=1000" "PAG_SEG_LEN=16000" "PAG_POST_INTERLEAVE=8" "PAG_POST_INTERLEAVE=32" "X=1"; do
  env $v python bench.py --steps 10 --warmup 1 --no-live-traffic --no-file-to-file --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
r = json.loads(sys.stdin.readlines()[-1]); c = r['config']
print('$v', 'ms_per_step', round(r['ms_per_step'], 1), 'walks', round(c['ms_walks_wall'], 1), 'successor stage', round(c['ms_successor_stage_wall'], 1), 'jobs', c['walk_jobs'], 'classifications', c['walk_classifications'], c['path_checksum'])" | tee -a $out
done
