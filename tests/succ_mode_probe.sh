#!/bin/bash
# GPU box: the successor stage at BASELINE configs[1] by the way its records are built (PAG_SUCC_MODE: two passes / fused)
out=${1:-gpurun_out/succ_mode_probe.txt}
: > $out
for m in twopass fused twopass fused; do
  line=$(PAG_SUCC_MODE=$m timeout 600 python bench.py --steps ${STEPS:-8} --warmup 2 --no-cpu-baseline --no-file-to-file --no-live-traffic 2>/dev/null | tail -1)
  echo "PAG_SUCC_MODE=$m $(python - "$line" <<'PY'
import json, sys
d = json.loads(sys.argv[1])
c = d["config"]
print(f"ms_per_step={d['ms_per_step']:.1f} ms_successor_stage_wall={c['ms_successor_stage_wall']:.1f} ms_walks_wall={c['ms_walks_wall']:.1f} checksum={c['path_checksum']}")
PY
)" | tee -a $out
done
