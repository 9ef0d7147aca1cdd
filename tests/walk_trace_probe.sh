#!/bin/bash
# GPU box: a quiet trace of the walks of one configs[1] block (PAG_WALK_TRACE), read by tests/walk_trace.py
mkdir -p gpurun_out
PAG_WALK_TRACE=1 python bench.py --steps 6 --warmup 1 --no-cpu-baseline --no-file-to-file --no-live-traffic > /dev/null 2> gpurun_out/walk_trace.log
python tests/walk_trace.py gpurun_out/walk_trace.log 1024 | cut -c1-220
gzip -9 -f gpurun_out/walk_trace.log
