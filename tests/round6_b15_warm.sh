#!/bin/bash
# Round 6: the steady-state step (warm pools, bench.py's schedule) of a block of configs[3]'s block-15 size: 90 Mb at 30x through one handle
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
PAGRAPH_TIMING=1 timeout 900 python bench.py --reads 270000 --ref-len 90000000 --steps 3 --warmup 1 --no-cpu-baseline --no-file-to-file --no-live-traffic > gpurun_out/r06_b15_warm.json 2> gpurun_out/r06_b15_warm.err
python - <<'PY' | tee gpurun_out/r06_b15_warm.txt
import json
d = json.loads(open('gpurun_out/r06_b15_warm.json').read().strip().split('\n')[-1]); c = d['config']
print(f"90 Mb at 30x (270 000 x 10 kb reads, {c['read_bases_per_gpu']} bases, {c['contigs']} contigs, {c['vertices']} vertices), warm pools: ms_per_step={d['ms_per_step']:.1f} value={d['value']:.3e} prepare={c['ms_prepare_wall']:.1f} build={c['ms_build_device']:.1f} (extract {c['ms_extract']:.1f} sort {c['ms_sort']:.1f} cluster {c['ms_cluster']:.1f} edges {c['ms_edges']:.1f}) succ={c['ms_successor_stage_wall']:.1f} walks={c['ms_walks_wall']:.1f} wait_host_half={c['ms_wait_for_previous_host_half']:.1f}")
PY
grep "traversal graph:" gpurun_out/r06_b15_warm.err | tail -1 | tee -a gpurun_out/r06_b15_warm.txt
grep "pag_travel laps" gpurun_out/r06_b15_warm.err | tail -1 | cut -c1-500 | tee -a gpurun_out/r06_b15_warm.txt
