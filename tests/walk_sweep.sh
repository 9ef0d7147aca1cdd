mkdir -p gpurun_out/r04h
run() { name=$1; shift; env "$@" python bench.py --steps 6 --warmup 1 --no-cpu-baseline --no-file-to-file 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['config']
print('$name', round(d['ms_per_step'],1), 'walks', round(c['ms_walks_wall'],1), 'succ', round(c['ms_successor_stage_wall'],1), 'wait', round(c['ms_wait_for_previous_host_half'],1), c['path_checksum'])
" >> gpurun_out/r04h/sweep.txt; }
run base A=1
run leap3000 PAG_LEAP_SEG_LEN=3000
run leap4000 PAG_LEAP_SEG_LEN=4000
run enddiv4 PAG_LEAP_END_DIV=4
run leap3000_enddiv4 PAG_LEAP_SEG_LEN=3000 PAG_LEAP_END_DIV=4
run seg8000 PAG_SEG_LEN=8000
run seg16000 PAG_SEG_LEN=16000
run host16 PAGH_OVERLAP_THREADS=16
run waves4 PAG_WALK_WAVES_PER_CU=4
cat gpurun_out/r04h/sweep.txt
