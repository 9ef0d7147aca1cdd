# One-off measurement (GPU box): the control thread's loop of the walks of configs[1], iteration by iteration (PAG_WALK_DEBUG).
mkdir -p gpurun_out/r04q
PAG_WALK_DEBUG=1 PAGRAPH_TIMING=1 python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-file-to-file > /dev/null 2> gpurun_out/r04q/loop.log
python - <<P
import re
lines=open("gpurun_out/r04q/loop.log").read().split("\n")
idx=[i for i,l in enumerate(lines) if "walker waves launched" in l]
lines=lines[idx[1]:]
prev=None
for ln in lines:
    m=re.match(r"\[walk\] t=([0-9.]+) ms loop: (.*)",ln)
    if not m: continue
    t=float(m.group(1))
    if t>80: print("%.1f (+%.1f) %s"%(t, t-(prev or t), m.group(2)))
    prev=t
P
grep "pag_travel laps\|pag_travel total" gpurun_out/r04q/loop.log | tail -2 | cut -c1-300
gzip -9 -f gpurun_out/r04q/loop.log
