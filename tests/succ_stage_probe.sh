#!/bin/bash
# GPU box: the successor stage alone under rocprofv3 (kernel summary), for the variants named in $VARIANTS (values of PAG_EMIT_EXP)
cd "$(dirname "$0")/.."
root=$PWD
mkdir -p gpurun_out
for v in ${VARIANTS:-0}; do
  cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/prof_s
  PAG_EMIT_EXP=$v rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_s -- python $root/tests/succ_stage_bench.py $ARGS > $root/gpurun_out/succ_stage_$v.log 2>&1
  f=$(find /tmp/prof_s -name "*kernel_stats.csv" | head -1)
  echo "== PAG_EMIT_EXP=$v $ARGS"; tail -2 $root/gpurun_out/succ_stage_$v.log
  python - $f <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    if any(x in r['Name'] for x in ('k_succ', 'k_order', 'k_view', 'k_compact', 'k_code', 'sort_scatter', 'sort_hist', 'k_edge', 'k_zone')):
        print(f"  {r['Name'][:60]:60s} calls={r['Calls']:>4s} avg_ms={float(r['AverageNs'])/1e6:8.3f} total_ms={float(r['TotalDurationNs'])/1e6:9.2f}")
PY
  cd $root
done
