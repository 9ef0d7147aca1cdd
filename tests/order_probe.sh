# One-off measurement (GPU box): k_order_apply with its random half in 1, 4, 8, 16 slices of the vertex id range.
root=$PWD; cd /tmp && export TMPDIR=/tmp
for sl in 1 4 8 16; do
rm -rf /tmp/prof
PAG_ORDER_SLICES=$sl rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -- python $root/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-file-to-file > /tmp/line.json 2>/dev/null
f=$(find /tmp/prof -name "*kernel_stats.csv" | head -1)
echo "slices $sl: $(grep k_order_apply $f | awk -F'",' '{print $2}')  $(python -c "
import json;d=json.loads(open('/tmp/line.json').read().strip().splitlines()[-1]);print('succ', round(d['config']['ms_successor_stage_wall'],1), d['config']['path_checksum'])")"
done
