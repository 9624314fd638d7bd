import os, sys, time, json
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR","127.0.0.1"); os.environ.setdefault("MASTER_PORT","29544")
dev=torch.device("cuda",0); torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
import intronprospector_amd as ipa
from intronprospector_amd import synth, distributed, _lib as L
import numpy as np
batch, names, _ = synth.short_reads(10_000_000, seed=synth.SEED_S)
keys=("tid","pos","flag","xs","nh","cigar_off","n_cigar","cigar")
arrs=[torch.from_numpy(batch[k]).to(dev) for k in keys]; ptrs=[t.data_ptr() for t in arrs]
n=len(batch["tid"]); nw=len(batch["cigar"])
ex=ipa.JunctionsExtractor(targets=names, device=0); ex.reserve(0,0,nw//2+1024)
X=distributed.RecordExchanger(dev, comm_device=dev)
# monkeypatch timing into run by re-implementing its steps
def timed_run():
    T={}
    def lap(name, t0):
        torch.cuda.synchronize(); T[name]=T.get(name,0)+time.perf_counter()-t0; return time.perf_counter()
    t=time.perf_counter()
    ex.clear(); ex.push_device(n,nw,ptrs,base_read_index=0); t=lap("clear+push",t)
    world=1; rank=0
    n_local=ex.records_pending()
    smp=ex.records_sample(X.N_SAMPLES).astype(np.int64); t=lap("sample",t)
    tt=torch.from_numpy(smp).to(dev); bufs=[torch.empty_like(tt)]; dist.all_gather(bufs,tt); allk=torch.cat(bufs); allk=torch.sort(allk[allk>=0]).values
    splitters=np.full(0, 0, np.uint64); t=lap("splitters",t)
    send=torch.empty(max(n_local,1)*32,dtype=torch.uint8,device=dev); t=lap("alloc_send",t)
    counts=ex.records_partition(splitters, send.data_ptr()); t=lap("partition",t)
    c=torch.from_numpy(counts.astype(np.int64)).to(dev); cm=[torch.empty_like(c)]; dist.all_gather(cm,c); rc=torch.stack(cm)[:,0].cpu().tolist(); t=lap("counts",t)
    n_recv=sum(rc); recv=torch.empty(max(n_recv,1)*32,dtype=torch.uint8,device=dev); t=lap("alloc_recv",t)
    dist.all_to_all_single(recv[:n_recv*32], send[:n_local*32], output_split_sizes=[x*32 for x in rc], input_split_sizes=[int(x)*32 for x in counts.tolist()]); t=lap("a2a",t)
    ex.records_adopt(n_recv, recv.data_ptr(), n); t=lap("adopt",t)
    rcx, table = ex.finalize_raw(L.IPX_MEM_DEVICE); t=lap("finalize",t)
    nj=int(table.n_junctions); co=torch.empty(nj,dtype=torch.int64,device=dev); ex._check(ex.lib.ipx_copy(ex.h, co.data_ptr(), table.creator_order, nj*8)); t=lap("copy_co",t)
    parts=distributed._all_gather_var(co); all_co=torch.sort(torch.cat(parts)).values; ij=torch.searchsorted(all_co, co); t=lap("ijunc",t)
    return T
for _ in range(3): timed_run()
acc={}
for _ in range(10):
    for k,v in timed_run().items(): acc[k]=acc.get(k,0)+v
print(json.dumps({k: round(v/10*1e3,3) for k,v in acc.items()}), "total_ms", round(sum(acc.values())/10*1e3,3))
dist.destroy_process_group()
