cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p -o s -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline > /tmp/b.json 2>/dev/null
python3 -c "
import csv
for r in list(csv.DictReader(open('/tmp/p/s_kernel_stats.csv')))[:14]: print(r['Name'][:60].ljust(60), r['Calls'].rjust(5), '%9.3f'%(float(r['AverageNs'])/1e6))
"
tail -1 /tmp/b.json | python3 -c "
import json,sys;d=json.loads(sys.stdin.read());print(d['value'],d['ms_per_step'],{k:round(v,1) for k,v in d['config'].items() if k.startswith('ms_')})"
