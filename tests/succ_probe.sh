# One-off measurement (GPU box): the successor stage of configs[1] — bench line + rocprofv3 durations of its kernels.
mkdir -p gpurun_out/r04p
bash tests/walk_tail_probe.sh 2>&1 | head -1
root=$PWD; cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/prof
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -- python $root/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-file-to-file > /dev/null 2>&1
f=$(find /tmp/prof -name "*kernel_stats.csv" | head -1)
grep "k_succ\|k_order_apply\|k_compact\|k_view" $f | awk -F'",' '{print substr($1,1,45), $2}'
