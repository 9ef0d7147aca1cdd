#!/bin/bash
# Round 6: the steady-state step (warm pools, bench.py's schedule) at 20x / 30x / 40x coverage of a 62.5 Mb block beside the 20x headline
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
: > gpurun_out/r06_coverage.txt
for cov in 20 30 40; do
  reads=$((62500000 / 10000 * cov))
  PAGRAPH_TIMING=1 timeout 900 python bench.py --reads $reads --ref-len 62500000 --steps 4 --warmup 1 --no-cpu-baseline --no-file-to-file --no-live-traffic > gpurun_out/r06_cov$cov.json 2> gpurun_out/r06_cov$cov.err
  python - $cov gpurun_out/r06_cov$cov.json <<'PY' | tee -a gpurun_out/r06_coverage.txt
import json, sys
d = json.loads(open(sys.argv[2]).read().strip().split('\n')[-1]); c = d['config']
print(f"coverage {sys.argv[1]}x of 62.5 Mb: bases {c.get('read_bases')} ms_per_step={d['ms_per_step']:.1f} value={d['value']:.3e} prepare={c['ms_prepare_wall']:.1f} build={c['ms_build_device']:.1f} (extract {c['ms_extract']:.1f} sort {c['ms_sort']:.1f} cluster {c['ms_cluster']:.1f} edges {c['ms_edges']:.1f}) succ={c['ms_successor_stage_wall']:.1f} walks={c['ms_walks_wall']:.1f} wait_host_half={c['ms_wait_for_previous_host_half']:.1f} view={c.get('view')}")
PY
  grep "traversal graph:" gpurun_out/r06_cov$cov.err | tail -1 | tee -a gpurun_out/r06_coverage.txt
done
