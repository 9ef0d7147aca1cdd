#!/bin/bash
# GPU box: the walks' wall time at BASELINE configs[1] against the number of walker waves per compute unit, for the two register
# budgets of k_walk_persistent (build/variants/libpagraph_hip_eu{1,2}.so: make [WALK_EU=2]; eu1 = one wave per SIMD at most).
# usage: tests/walk_waves_sweep.sh OUT.txt
out=${1:-gpurun_out/walk_waves_sweep.txt}
: > $out
keep=/tmp/lib_keep.so
cp aligngraph2_amd/libpagraph_hip.so $keep
for v in eu1 eu2; do
  [ -f build/variants/libpagraph_hip_$v.so ] || continue
  cp build/variants/libpagraph_hip_$v.so aligngraph2_amd/libpagraph_hip.so
  for w in ${WAVES:-3 4 5 6}; do
    line=$(PAG_WALK_WAVES_PER_CU=$w timeout 600 python bench.py --steps ${STEPS:-6} --warmup 2 --no-cpu-baseline --no-file-to-file 2>/dev/null | tail -1)
    echo "$v waves_per_cu=$w $(python - "$line" <<'PY'
import json, sys
d = json.loads(sys.argv[1])
c = d["config"]
print(f"ms_per_step={d['ms_per_step']:.1f} ms_walks_wall={c['ms_walks_wall']:.1f} ms_successor_stage_wall={c['ms_successor_stage_wall']:.1f} ms_build={c['ms_build_device']:.1f} checksum={c['path_checksum']}")
PY
)" | tee -a $out
  done
done
cp $keep aligngraph2_amd/libpagraph_hip.so
