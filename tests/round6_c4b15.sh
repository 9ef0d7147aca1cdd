#!/bin/bash
cd "$(dirname "$0")/.."
export PAGRAPH_TIMING=1
timeout 900 python tests/c4_blocks.py gpurun_out/r06_c4_b15.json --blocks 15 --one-gpu-bases 3.0e9 --cross-check 0 > gpurun_out/r06_c4_b15.log 2>&1
grep -v "^\[timing\] \(stitch\|leaping\|last rounds\|pieces\|walks redone\|segment jobs\|buildPath\|assemble\)" gpurun_out/r06_c4_b15.log | tail -40 | cut -c1-400
