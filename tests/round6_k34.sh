#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_build_parity.py tests/test_gpu_configs.py -x -q 2>&1 | tail -3
python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-file-to-file --no-live-traffic 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().split('\n')[-1]); c=d['config']
print('configs[1]: ms_per_step=%.1f build=%.1f (extract %.1f sort %.1f cluster %.1f edges %.1f) succ=%.1f walks=%.1f' % (d['ms_per_step'], c['ms_build_device'], c['ms_extract'], c['ms_sort'], c['ms_cluster'], c['ms_edges'], c['ms_successor_stage_wall'], c['ms_walks_wall']))"
tests/round6_coverage.sh 2>&1 | grep "coverage"
PAGRAPH_TIMING=1 timeout 1500 python -m pytest tests/test_gpu_at_size.py -x -q -s 2>&1 | grep -v "^\[timing\] \(stitch\|leaping\|last rounds\|pieces\|walks redone\|segment jobs\|buildPath\|assemble\)" | tail -30 | cut -c1-300
