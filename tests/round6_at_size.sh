#!/bin/bash
# Round 6: where the seconds and the bytes go at the sizes of BASELINE configs[2] / configs[3] (one GPU box).
#   tests/round6_at_size.sh [bench] [c4] [c3]
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export PAGRAPH_TIMING=1
for what in "$@"; do
  case $what in
    bench) python bench.py > gpurun_out/r06_bench.log 2> gpurun_out/r06_bench.err; tail -c 3000 gpurun_out/r06_bench.log ;;
    c4) timeout 900 python tests/c4_blocks.py gpurun_out/r06_c4_b15_b20.json --blocks "${C4_BLOCKS:-20,15}" --one-gpu-bases 3.0e9 --cross-check 0 > gpurun_out/r06_c4.log 2>&1; tail -n 60 gpurun_out/r06_c4.log ;;
    c3) timeout 1500 python tests/c3_rank_serial.py gpurun_out/r06_c3.json --ranks 4 > gpurun_out/r06_c3.log 2>&1; tail -n 80 gpurun_out/r06_c3.log ;;
  esac
done
