# Round-end evidence (GPU box), one command: the full GPU suite, the default bench line, the rocprofv3 kernel summary of the bench.
# Everything lands under gpurun_out/evidence/; copy what is to be judged into profiles/.
out=gpurun_out/evidence
mkdir -p $out
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $out/smoke.log 2>&1; tail -1 $out/smoke.log
python -m pytest tests -q -m gpu -x > $out/gputest_full_suite.log 2>&1
tail -3 $out/gputest_full_suite.log
python bench.py > $out/bench_line.json 2> $out/bench_stderr.log
tail -c 600 $out/bench_line.json
root=$PWD
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -- python $root/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-file-to-file > $root/$out/bench_line_under_rocprof.json 2>/dev/null
f=$(find /tmp/prof -name "*kernel_stats.csv" | head -1)
grep -v "at::native\|at::cuda" $f | head -60 > $root/$out/kernel_stats.csv
cd $root
