# One-off measurement (GPU box): the walks of BASELINE configs[1] under a few settings of the control thread / the pieces.
mkdir -p gpurun_out/r04h
run() { name=$1; shift; env "$@" python bench.py --steps 6 --warmup 1 --no-cpu-baseline --no-file-to-file 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['config']
print('$name', round(d['ms_per_step'],1), 'walks', round(c['ms_walks_wall'],1), 'succ', round(c['ms_successor_stage_wall'],1), 'wait', round(c['ms_wait_for_previous_host_half'],1), c['path_checksum'])
" >> gpurun_out/r04h/sweep2.txt; }
rm -f gpurun_out/r04h/sweep2.txt
run base A=1
run group8 PAG_POST_GROUP=8
run group12 PAG_POST_GROUP=12
run group16 PAG_POST_GROUP=16
run group24 PAG_POST_GROUP=24
run group32 PAG_POST_GROUP=32
run base2 A=1
cat gpurun_out/r04h/sweep2.txt
