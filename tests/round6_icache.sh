#!/bin/bash
# Round 6: do the large kernels (walker 100 KB, emit 29 KB) miss the instruction cache?  SQC / SQ instruction-fetch counters, one pass
cd "$(dirname "$0")/.."
root=$PWD
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -i "icache\|ifetch\|INST_CACHE" | sort -u | head -30 > $root/gpurun_out/r06_icache_counters.txt
cat $root/gpurun_out/r06_icache_counters.txt | cut -c1-160
for set in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES" "SQ_IFETCH SQ_WAIT_IFETCH SQ_WAVE_CYCLES"; do
  rm -rf /tmp/pmc_ic
  PAG_WALK_IDLE_S=5 timeout 600 rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/pmc_ic -o c -- python $root/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-file-to-file --no-live-traffic > /dev/null 2>&1
  python - "$set" <<'PY' | tee -a $root/gpurun_out/r06_icache_probe.txt
import collections, csv, glob, sys
acc = collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob('/tmp/pmc_ic/**/c_counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'pagdev' in r['Kernel_Name']:
            acc[r['Kernel_Name'].split('(')[0].replace('void ', '')][r['Counter_Name']] += float(r['Counter_Value'])
print('==', sys.argv[1])
for k in sorted(acc, key=lambda k: -max(acc[k].values())):
    if max(acc[k].values()) > 1e6:
        print(f"{k[:44]:44s}", {c: f"{v:.3e}" for c, v in acc[k].items()})
PY
done
