# development aid: the round's records for profiles/ (bench line, kernel stats, PMC traffic, k sweep)
mkdir -p gpurun_out/r3p
O=$GRAFT_REPO_ROOT/gpurun_out/r3p
python bench.py --file-to-file > $O/r03_bench_c2_line.json 2> $O/bench.err; tail -c 600 $O/r03_bench_c2_line.json
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/r03_bench_c2_line_under_rocprof.json 2> $O/prof.err
python - <<PY
import csv,glob
f=glob.glob('$O/prof/**/*kernel_stats.csv',recursive=True)[0]
rows=[r for r in csv.DictReader(open(f)) if 'pagdev' in r['Name'] or 'rocclr' in r['Name']]
w=csv.DictWriter(open('$O/r03_bench_c2_kernel_stats.csv','w'),fieldnames=list(rows[0].keys())); w.writeheader(); w.writerows(rows)
print(len(rows),'kernels')
PY
timeout 1500 python $GRAFT_REPO_ROOT/tests/pmc_traffic.py $O/r03_pmc_hbm_traffic.json > $O/pmc.log 2>&1; tail -12 $O/pmc.log
timeout 2400 python $GRAFT_REPO_ROOT/tests/k_sweep.py $O/r03_k_sweep_sort.json > $O/ksweep.log 2>&1; tail -4 $O/ksweep.log | cut -c1-400
rm -rf $O/prof
