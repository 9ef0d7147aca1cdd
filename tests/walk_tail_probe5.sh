mkdir -p gpurun_out/r05t
PAG_WALK_DEBUG=1 PAGRAPH_TIMING=1 python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-file-to-file > /dev/null 2> gpurun_out/r05t/walk.log
grep "round . over\|round [23]:" gpurun_out/r05t/walk.log | tail -57 | grep -v "leap$" | cut -c1-260
python tests/walk_timeline.py gpurun_out/r05t/walk.log 1024 | head -16 | cut -c1-400
grep -c . gpurun_out/r05t/walk.log
gzip -9 -f gpurun_out/r05t/walk.log
