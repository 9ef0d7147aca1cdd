#!/bin/bash
# GPU box, round 5: the PMC records kept under profiles/ — HBM bytes per launch of the build / successor-stage kernels
# (tests/pmc_traffic.py), SQ counters incl. the walker's (tests/pmc_kernel_mix.py), and the whole configs[1] workload as text
# through the compiled reference (-t 64) and the drop-in (tests/c2_text_runs.py: bench.py's cached full_workload record).
root=$PWD
mkdir -p $root/gpurun_out
cd /tmp && export TMPDIR=/tmp
timeout 900 python $root/tests/pmc_traffic.py $root/gpurun_out/r05_pmc_hbm_traffic.json > $root/gpurun_out/r05_pmc_hbm_traffic.txt 2>&1
tail -25 $root/gpurun_out/r05_pmc_hbm_traffic.txt
timeout 1500 python $root/tests/pmc_kernel_mix.py $root/gpurun_out/r05_pmc_kernel_mix.json > $root/gpurun_out/r05_pmc_kernel_mix.txt 2>&1
grep "k_walk\|k_succ\|sort_scatter<7\|extract_kernel<true" $root/gpurun_out/r05_pmc_kernel_mix.txt | cut -c1-700
cd $root
if [ "$1" = "with-reference" ]; then
  timeout 1200 python tests/c2_text_runs.py gpurun_out/r05_c2_text_runs.json > gpurun_out/r05_c2_text_runs.txt 2>&1
  tail -5 gpurun_out/r05_c2_text_runs.txt | cut -c1-600
fi
