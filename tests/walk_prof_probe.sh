#!/bin/bash
# GPU box: section timers of the chain jobs at the end of the walks (make WALK_PROF=1 build in build/variants/libpagraph_hip_prof.so)
mkdir -p gpurun_out/r05t
keep=/tmp/lib_keep.so
cp aligngraph2_amd/libpagraph_hip.so $keep
cp build/variants/libpagraph_hip_prof.so aligngraph2_amd/libpagraph_hip.so
PAG_WALK_DEBUG=1 PAGRAPH_TIMING=1 python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-file-to-file > /dev/null 2> gpurun_out/r05t/walk_prof.log
cp $keep aligngraph2_amd/libpagraph_hip.so
awk '/walker waves launched/{n++} n>=2' gpurun_out/r05t/walk_prof.log | grep -A1 "chain . job done" | paste - - - | awk '{split($5,a,"\\.\\."); d=a[2]-a[1]; print d, $0}' | sort -rn | head -14 | cut -c1-900
python tests/walk_timeline.py gpurun_out/r05t/walk_prof.log 1024 | head -3
gzip -9 -f gpurun_out/r05t/walk_prof.log
