#!/bin/bash
cd "$(dirname "$0")/.."
timeout 900 python -m pytest tests/test_gpu_succ_modes.py tests/test_gpu_succ_golden.py -x -q 2>&1 | tail -3
for b in 6 8 4; do PAG_EMIT_BLOCKS=$b VARIANTS="0" ARGS="--reps 3" tests/succ_stage_probe.sh 2>&1 | grep "==\|k_succ\|rep 2" | sed "s/^/blocks=$b /"; done | tee gpurun_out/r06_emit_exp2.txt
