import sys, numpy as np
sys.path.insert(0,'/root/repo')
from intronprospector_amd import synth
from tests.oracle_binding import Oracle

def load_keys():
    b,names,_=synth.short_reads(10_000_000, seed=20251024)
    o=Oracle(targets=names); o.process(b); tab=o.table(); o.close()
    n=tab['n_offsets'].astype(np.int64); offs=tab['offsets'].astype(np.uint64)
    j=np.repeat(np.arange(len(n),dtype=np.uint64), n)
    return (j<<np.uint64(16))|offs

def splitters(keys, leaf_avg=1365, oversample=16, heavy_d=10, listcap=4096):
    n=len(keys)
    k_target=min(max(1,-(-n//leaf_avg)),43000)
    m2=min(n,k_target*oversample)
    k1=max(-(-m2//1800), -(-k_target//170)); k1=min(max(k1,1),256)
    m1=min(max(16*k1,256),4096)
    m=min(n,m1)
    s1=np.sort(keys[(np.arange(m,dtype=np.int64)*n)//m])
    coarse=s1[((np.arange(1,k1,dtype=np.int64))*m)//k1]
    g=np.arange(m2,dtype=np.int64)
    s2=keys[(g*n+n//2)//m2]
    c=np.searchsorted(coarse,s2,side='right')
    allsp=[]
    dropped=0
    for r in range(k1):
        lst=np.sort(s2[c==r])[:listcap]   # (order of arrival differs on the device when > listcap; ignore)
        mm=len(lst)
        f=(mm+oversample//2)//oversample; f=min(max(f,1),256)
        out=[]
        for t in range(f-1):
            rr=((t+1)*mm)//f
            v=lst[rr]
            dup = t>=1 and v==lst[(t*mm)//f]
            heavy = (rr+heavy_d<mm and lst[rr+heavy_d]==v) or (rr>=heavy_d and lst[rr-heavy_d]==v)
            if dup: continue
            out.append(v)
            if heavy: out.append(v+np.uint64(1))
        if len(out)>255: dropped+=len(out)-255; out=out[:255]
        allsp.append((r,np.array(out,dtype=np.uint64)))
    return coarse, allsp, dropped

def leaf_sizes(keys, coarse, allsp):
    # leaf id = (coarse idx, fine idx) ; build global boundary list
    bounds=[]
    for r,sp in allsp:
        if r>0: bounds.append(coarse[r-1:r])
        bounds.append(sp)
    bnd=np.concatenate(bounds)
    # boundaries must be ascending (fine splitters lie inside their coarse range)
    ids=np.searchsorted(bnd,keys,side='right')
    cnt=np.bincount(ids,minlength=len(bnd)+1)
    return ids,cnt

if __name__=="__main__":
    keys=load_keys()
    np.save('/tmp/sim/keys.npy',keys)
