#!/bin/bash
# Round 6: (a) the compiled reference at MATCHED width (-t 16, its own threads, the box's 16-CPU quota) on the configs[1] text files,
# (b) the full-size oracle comparison of every CSR array + the full-size host-walk comparison (PAG_C2_FULL=1)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout 1500 python tests/c2_text_runs.py gpurun_out/r06_c2_text_runs_t16.json --ref-threads 16 > gpurun_out/r06_c2_text_runs_t16.log 2>&1; tail -3 gpurun_out/r06_c2_text_runs_t16.log | cut -c1-400
( time PAG_C2_FULL=1 timeout 2400 python -m pytest tests/test_gpu_configs.py -x -q -s -m gpu -k "config1" ) > gpurun_out/r06_c2_full_parity.log 2>&1; tail -12 gpurun_out/r06_c2_full_parity.log
