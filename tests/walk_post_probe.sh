#!/bin/bash
# GPU box: the walks' wall time at BASELINE configs[1] against how the first rounds' segment jobs enter the ring: equal shares per
# contig and turn (PAG_POST_PROPORTIONAL=0, until round 5) / in proportion to a contig's jobs, interleave widths, and the
# longest contigs through the ring early (PAG_POST_SPREAD).   usage: tests/walk_post_probe.sh OUT.txt
out=${1:-gpurun_out/walk_post_probe.txt}
: > $out
run() {
  line=$(env "$@" timeout 600 python bench.py --steps ${STEPS:-10} --warmup 2 --no-cpu-baseline --no-file-to-file 2>/dev/null | tail -1)
  echo "$* $(python - "$line" <<'PY'
import json, sys
d = json.loads(sys.argv[1])
c = d["config"]
print(f"ms_per_step={d['ms_per_step']:.1f} ms_walks_wall={c['ms_walks_wall']:.1f} ms_successor_stage_wall={c['ms_successor_stage_wall']:.1f} checksum={c['path_checksum']}")
PY
)" | tee -a $out
}
for il in 8 16 32; do run PAG_POST_PROPORTIONAL=0 PAG_POST_INTERLEAVE=$il; done
for il in 8 16 32 64; do for sp in 1.0 0.7 0.5; do run PAG_POST_PROPORTIONAL=1 PAG_POST_INTERLEAVE=$il PAG_POST_SPREAD=$sp; done; done
