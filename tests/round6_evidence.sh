#!/bin/bash
# Round-6 evidence (GPU box), one command: smoke, the full GPU suite with durations, the default bench line, the bench at the driver's
# settings, the rocprofv3 kernel summary of the bench, HBM traffic of the build / successor-stage kernels (PMC passes).
# Everything lands under gpurun_out/evidence6/; copy what is to be judged into profiles/.
cd "$(dirname "$0")/.."
out=gpurun_out/evidence6
mkdir -p $out
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $out/smoke.log 2>&1; tail -1 $out/smoke.log
t0=$(date +%s)
timeout 2400 python -m pytest tests -q -m gpu -x --durations=12 > $out/gputest_full_suite.log 2>&1
echo "suite wall: $(( $(date +%s) - t0 )) s" >> $out/gputest_full_suite.log
tail -4 $out/gputest_full_suite.log
python bench.py > $out/bench_line.json 2> $out/bench_stderr.log
python bench.py --steps 20 --warmup 5 > $out/bench_line_steps20.json 2> $out/bench_stderr_steps20.log
python - <<'PY'
import json
for f in ('bench_line.json', 'bench_line_steps20.json'):
    d = json.loads(open('gpurun_out/evidence6/' + f).read().strip().split('\n')[-1]); c = d['config']
    print(f, 'ms_per_step=%.1f value=%.3e frac=%.4f per_pass=%.3f' % (d['ms_per_step'], d['value'], d['roofline']['frac'], d['roofline']['per_pass']['frac']),
          'prepare=%.1f build=%.1f (extract %.1f sort %.1f cluster %.1f edges %.1f) succ=%.1f walks=%.1f' % (c['ms_prepare_wall'], c['ms_build_device'], c['ms_extract'], c['ms_sort'], c['ms_cluster'], c['ms_edges'], c['ms_successor_stage_wall'], c['ms_walks_wall']))
PY
root=$PWD
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -- python $root/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-file-to-file --no-live-traffic > $root/$out/bench_line_under_rocprof.json 2>/dev/null
f=$(find /tmp/prof -name "*kernel_stats.csv" | head -1)
grep -v "at::native\|at::cuda" $f | head -70 > $root/$out/kernel_stats.csv
timeout 900 python $root/tests/pmc_traffic.py $root/$out/pmc_hbm_traffic.json > $root/$out/pmc_hbm_traffic.txt 2>&1
cd $root
head -12 $out/kernel_stats.csv | cut -c1-150
cat $out/pmc_hbm_traffic.txt | tail -30
