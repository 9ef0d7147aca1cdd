import glob,os
tot=0;z=0;ztail=0;tail=0;files=0;with_zero_tail=0
for f in glob.glob('/dev/shm/walkcheck_*/0_*.txt'):
    L=open(f).read().splitlines()[1:]
    c=[int(l.split('\t')[0].split(',')[1]) for l in L if l]
    n=len(c); files+=1
    t0=int(n*0.88)
    zz=sum(1 for x in c if x==0); zt=sum(1 for x in c[t0:] if x==0)
    tot+=n; z+=zz; tail+=n-t0; ztail+=zt; with_zero_tail+= zt>0
print('files',files,'vertices',tot,'ctg=0',z,'tail vertices',tail,'ctg=0 in tail',ztail,'files with zero in tail',with_zero_tail)
