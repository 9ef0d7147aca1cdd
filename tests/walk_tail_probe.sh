mkdir -p gpurun_out/r04g
python bench.py --steps 6 --warmup 1 --no-cpu-baseline --no-file-to-file 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['config']; print(d['ms_per_step'], 'walks', c['ms_walks_wall'], 'succ', c['ms_successor_stage_wall'], 'sort', c['ms_sort'], c['path_checksum'])"
PAG_WALK_DEBUG=1 PAGRAPH_TIMING=1 python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-file-to-file > /dev/null 2> gpurun_out/r04g/walk4.log
grep "round . over\|round [23]:" gpurun_out/r04g/walk4.log | tail -57 | grep -v "leap$" | cut -c1-230
python tests/walk_timeline.py gpurun_out/r04g/walk4.log 768 | head -14 | cut -c1-300
gzip -9 -f gpurun_out/r04g/walk4.log
