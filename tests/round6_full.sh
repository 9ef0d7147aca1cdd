#!/bin/bash
# Round 6 checkpoint on the GPU box: full GPU suite, default bench line, then the at-size runs given as arguments (c4, c3)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -q -m gpu -x --durations=15 > gpurun_out/r06_gputest.log 2>&1; tail -25 gpurun_out/r06_gputest.log
python bench.py > gpurun_out/r06_bench2.log 2> gpurun_out/r06_bench2.err; python - <<'PY'
import json
d = json.loads(open('gpurun_out/r06_bench2.log').read().strip().split('\n')[-1]); c = d['config']
print('bench: ms_per_step=%.1f value=%.3e succ=%.1f walks=%.1f build=%.1f' % (d['ms_per_step'], d['value'], c['ms_successor_stage_wall'], c['ms_walks_wall'], c['ms_build_device']))
PY
tests/round6_at_size.sh "$@"
