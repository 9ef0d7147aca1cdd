cd /tmp && export TMPDIR=/tmp
mkdir -p $GRAFT_REPO_ROOT/gpurun_out/final
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p -o s -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 > $GRAFT_REPO_ROOT/gpurun_out/final/stats_line.json 2>/dev/null
cp /tmp/p/s_kernel_stats.csv $GRAFT_REPO_ROOT/gpurun_out/final/kernel_stats.csv
cd $GRAFT_REPO_ROOT
python bench.py > gpurun_out/final/bench_default.json 2> gpurun_out/final/bench_default.err
tail -c 400 gpurun_out/final/bench_default.json
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
