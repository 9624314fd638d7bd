import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
import intronprospector_amd as ipa
from intronprospector_amd import synth
batch, names, _ = synth.short_reads(10_000_000, seed=synth.SEED_S)
dev = torch.device("cuda", 0)
keys=("tid","pos","flag","xs","nh","cigar_off","n_cigar","cigar")
arrs=[torch.from_numpy(batch[k]).to(dev) for k in keys]; ptrs=[t.data_ptr() for t in arrs]
n=len(batch["tid"]); nw=len(batch["cigar"])
ex=ipa.JunctionsExtractor(targets=names, device=0); ex.reserve(0,0,nw//2+1024)
ex.push_device(n,nw,ptrs,base_read_index=0)
R=ex.records_pending()
send=torch.empty(R*32,dtype=torch.uint8,device=dev)
for parts in (1,2,8,32):
    smp=np.sort(ex.records_sample(512).astype(np.uint64))
    spl=smp[(np.arange(1,parts)*512)//parts].astype(np.uint64)
    for _ in range(3): c=ex.records_partition(spl, send.data_ptr())
    torch.cuda.synchronize(); t0=time.perf_counter()
    for _ in range(10): c=ex.records_partition(spl, send.data_ptr())
    torch.cuda.synchronize(); dt=(time.perf_counter()-t0)/10
    print(parts, "parts: %.3f ms" % (dt*1e3), "counts", c.tolist()[:4], int(c.sum())==R)
