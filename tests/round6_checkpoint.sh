#!/bin/bash
# Round 6 checkpoint on the GPU box: full GPU suite with its durations (the driver's limit is 1 200 s), default bench line,
# rocprofv3 kernel summary of the bench.  Lands under gpurun_out/r06ck/.
cd "$(dirname "$0")/.."
out=gpurun_out/r06ck
mkdir -p $out
t0=$(date +%s)
timeout 2400 python -m pytest tests -q -m gpu -x --durations=25 > $out/gputest.log 2>&1; tail -40 $out/gputest.log
echo "suite wall: $(( $(date +%s) - t0 )) s" | tee -a $out/gputest.log
python bench.py > $out/bench_line.json 2> $out/bench_stderr.log
tail -c 3000 $out/bench_line.json
root=$PWD
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -- python $root/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-file-to-file > $root/$out/bench_line_under_rocprof.json 2>/dev/null
f=$(find /tmp/prof -name "*kernel_stats.csv" | head -1)
grep -v "at::native\|at::cuda" $f | head -70 > $root/$out/kernel_stats.csv
cd $root
head -40 $out/kernel_stats.csv
