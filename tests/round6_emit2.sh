#!/bin/bash
# Round 6: allocation probe; the emission kernel with the four edges' candidates requested together: parity tests, the stage under rocprofv3
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
tests/harness/bin/alloc_probe 32 2>&1 | tee gpurun_out/r06_alloc_probe.txt
timeout 900 python -m pytest tests/test_gpu_succ_modes.py tests/test_gpu_succ_golden.py -x -q 2>&1 | tail -3
ARGS="--reps 3" tests/succ_stage_probe.sh 2>&1 | tee gpurun_out/r06_emit2_stage.txt
