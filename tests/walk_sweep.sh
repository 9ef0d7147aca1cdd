# One-off measurement (GPU box): the walks of BASELINE configs[1] under a few settings of the control thread / the pieces.
mkdir -p gpurun_out/r04h
run() { name=$1; shift; env "$@" python bench.py --steps 6 --warmup 1 --no-cpu-baseline --no-file-to-file 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['config']
print('$name', round(d['ms_per_step'],1), 'walks', round(c['ms_walks_wall'],1), 'succ', round(c['ms_successor_stage_wall'],1), 'wait', round(c['ms_wait_for_previous_host_half'],1), c['path_checksum'])
" >> gpurun_out/r04h/sweep2.txt; }
rm -f gpurun_out/r04h/sweep2.txt
run base A=1
run host_tail PAG_DEVICE_TAIL=0
run no_kept_segments PAG_WALK_KEEP_SEGMENTS=0
run interleave0 PAG_POST_INTERLEAVE=0
run waves4 PAG_WALK_WAVES_PER_CU=4
run seg8k PAG_SEG_LEN=8000
run succ_heavy0 PAG_SUCC_HEAVY=0
run base2 A=1
cat gpurun_out/r04h/sweep2.txt
