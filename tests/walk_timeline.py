"""Development aid: busy walker waves per 10 ms and the last jobs to finish, from the stderr of a run with PAG_WALK_DEBUG=1
(the job lines carry the device clock at which a wave took / finished the job).
  PAG_WALK_DEBUG=1 PAGRAPH_TIMING=1 python bench.py --steps 2 --warmup 1 2> walk.log; python tests/walk_timeline.py walk.log [waves]"""
import re,sys
import numpy as np
txt=open(sys.argv[1]).read().splitlines()
idx=[i for i,l in enumerate(txt) if 'walker waves launched' in l][-1]
lines=txt[idx:]
W=int(sys.argv[2]) if len(sys.argv)>2 else 256
jobs=[]
for l in lines:
    m=re.search(r't=([\d.]+) ms dev ([\d.]+)\.\.([\d.]+) contig (\d+) (segment|chain) (\d+).*?(\+?)(\d+) vertices',l)
    if m: jobs.append((float(m[2]),float(m[3]),int(m[4]),m[5],int(m[6]),int(m[8]),float(m[1])))
t0=min(j[0] for j in jobs); t1=max(j[1] for j in jobs)
print(len(jobs),'span ms %.1f'%(t1-t0),'busy/W %.1f'%(sum(j[1]-j[0] for j in jobs)/W))
nb=int((t1-t0)/10)+1
u=np.zeros(nb)
for b,e,*_ in jobs:
    for k in range(int((b-t0)/10),int((e-t0)/10)+1):
        lo=max(b,t0+k*10);hi=min(e,t0+(k+1)*10)
        if hi>lo:u[k]+=hi-lo
print([int(x/10) for x in u])
jobs.sort(key=lambda j:j[1],reverse=True)
for j in jobs[:10]:print('%.1f..%.1f dur %.1f ctg %d %s %d verts %d host t=%.1f'%(j[0]-t0,j[1]-t0,j[1]-j[0],j[2],j[3],j[4],j[5],j[6]))
for l in lines:
    if 'pag_travel laps' in l or 'pag_travel total' in l: print(l[:330])
