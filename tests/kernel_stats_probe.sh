#!/bin/bash
# GPU box: rocprofv3 kernel summary of a short bench run -> gpurun_out/<name>_kernel_stats.csv (pagdev kernels, per-launch averages)
name=${1:-probe}
root=$PWD
mkdir -p $root/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_$name
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$name -- python $root/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-file-to-file > $root/gpurun_out/${name}_line_under_rocprof.json 2>/dev/null
f=$(find /tmp/prof_$name -name "*kernel_stats.csv" | head -1)
grep -v "at::native\|at::cuda" $f | head -70 > $root/gpurun_out/${name}_kernel_stats.csv
cd $root
python - $root/gpurun_out/${name}_kernel_stats.csv <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:48]:
    print(f"{r['Name'][:70]:70s} calls={r['Calls']:>5s} per_step_ms={float(r['TotalDurationNs'])/1e6/4:8.2f} avg_us={float(r['AverageNs'])/1e3:9.1f}")
PY
