#!/bin/bash
# GPU box: host threads of a block's host half while it runs beside the next block's device work (PAGH_OVERLAP_THREADS): the wait
# for the previous block's host half against the pool size
out=${1:-gpurun_out/overlap_threads_probe.txt}
: > $out
for v in ${PROBE_THREADS:-12 14 16 10 8 12 16}; do
  PAGH_OVERLAP_THREADS=$v python bench.py --steps 10 --warmup 1 --no-live-traffic --no-file-to-file --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
r = json.loads(sys.stdin.readlines()[-1]); c = r['config']
print('PAGH_OVERLAP_THREADS=$v', 'ms_per_step', round(r['ms_per_step'], 1), 'wait', round(c['ms_wait_for_previous_host_half'], 1), 'walks', round(c['ms_walks_wall'], 1), 'succ', round(c['ms_successor_stage_wall'], 1), 'build', round(c['ms_pag_process_wall'],1), 'host epilogue', round(c['ms_traverse_host_epilogue'], 1))" | tee -a $out
done
