# development aid: per-kernel times of the bench step (rocprofv3 --kernel-trace --stats)
mkdir -p gpurun_out/r3a/prof
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r3a/prof -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/r3a/prof/bench.out 2> $GRAFT_REPO_ROOT/gpurun_out/r3a/prof/bench.err
cd $GRAFT_REPO_ROOT
ls gpurun_out/r3a/prof | head
python - <<'PY'
import csv,glob
f=glob.glob('gpurun_out/r3a/prof/**/*kernel_stats.csv',recursive=True)
print(f)
rows=list(csv.DictReader(open(f[0])))
for r in rows[:40]:
    print(r['Name'][:90], r['Calls'], r['TotalDurationNs'], r['AverageNs'], r['Percentage'])
PY
