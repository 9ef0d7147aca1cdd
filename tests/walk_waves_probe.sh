# One-off measurement (GPU box): job timeline of the walks of configs[1] with 3 (default) and 6 walker waves per CU.
mkdir -p gpurun_out/r04n
for w in 3 6; do
PAG_WALK_WAVES_PER_CU=$w PAG_WALK_DEBUG=1 PAGRAPH_TIMING=1 python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-file-to-file > /dev/null 2> gpurun_out/r04n/walk_w$w.log
echo "waves per CU $w"
python tests/walk_timeline.py gpurun_out/r04n/walk_w$w.log $((w * 256)) | head -3 | cut -c1-300
grep "pag_travel laps\|pag_travel total\|walker waves launched" gpurun_out/r04n/walk_w$w.log | tail -3 | cut -c1-400
python - <<P
import re,statistics
d=[]
for ln in open("gpurun_out/r04n/walk_w$w.log"):
    m=re.search(r"dev ([0-9.]+)\.\.([0-9.]+) contig \d+ segment \d+ done: (\d+) vertices",ln)
    if m: d.append((float(m.group(2))-float(m.group(1)), int(m.group(3))))
d=d[len(d)//2:]
print("segment jobs (timed call):", len(d), "median duration ms", round(statistics.median(x for x,_ in d),2), "us per vertex", round(1e3*sum(x for x,_ in d)/sum(v for _,v in d),2))
P
gzip -9 -f gpurun_out/r04n/walk_w$w.log
done
