#!/bin/bash
# Round 6: the successor stage rebuilt (k_succ_emit -> sort -> k_succ_finish): parity tests, then the step with 4 / 6 / 8 emit
# blocks per CU and the rocprofv3 kernel summary.
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_succ_modes.py tests/test_gpu_succ_golden.py -x -q > gpurun_out/r06_succ_tests.log 2>&1; tail -5 gpurun_out/r06_succ_tests.log
for b in 6 8 4; do
  PAG_EMIT_BLOCKS=$b python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-file-to-file 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().split('\n')[-1]); c=d['config']
print('emit_blocks=$b', 'ms_per_step=%.1f' % d['ms_per_step'], 'succ=%.1f' % c['ms_successor_stage_wall'], 'walks=%.1f' % c['ms_walks_wall'], 'checksum', c.get('traversal_checksum', c.get('path_checksum')))
" | tee -a gpurun_out/r06_succ_probe.txt
done
tests/kernel_stats_probe.sh r06_succ | tee gpurun_out/r06_succ_kernels.txt | head -40
