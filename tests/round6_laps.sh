#!/bin/bash
# Round 6: the stage laps of every step of the default bench (PAGRAPH_TIMING), then the at-size runs given as arguments (c4, c3)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
PAGRAPH_TIMING=1 python bench.py --steps 6 --warmup 1 --no-cpu-baseline --no-file-to-file --no-live-traffic > gpurun_out/r06_laps_bench.json 2> gpurun_out/r06_laps_bench.err
grep "traversal graph:\|pag_travel laps\|pag_process wall" gpurun_out/r06_laps_bench.err | cut -c1-700
python - <<'PY'
import json
d = json.loads(open('gpurun_out/r06_laps_bench.json').read().strip().split('\n')[-1]); c = d['config']
print('bench: ms_per_step=%.1f succ=%.1f walks=%.1f build=%.1f prepare=%.1f' % (d['ms_per_step'], c['ms_successor_stage_wall'], c['ms_walks_wall'], c['ms_build_device'], c['ms_prepare_wall']))
PY
tests/round6_at_size.sh "$@"
