"""Where the host's time goes in a step of the slotted exchange with one rank (cProfile over 200 steps)."""
import os, sys, time, cProfile, pstats, io
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29545")
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
import intronprospector_amd as ipa
from intronprospector_amd import synth, distributed
batch, names, _ = synth.short_reads(10_000_000, seed=synth.SEED_S)
keys = ("tid", "pos", "flag", "xs", "nh", "cigar_off", "n_cigar", "cigar")
arrs = [torch.from_numpy(batch[k]).to(dev) for k in keys]; ptrs = [t.data_ptr() for t in arrs]
n = len(batch["tid"]); nw = len(batch["cigar"])
ex = ipa.JunctionsExtractor(targets=names, device=0); ex.reserve(0, 0, nw // 2 + 1024)
X = distributed.RecordExchanger(dev, comm_device=dev); X.strict = False
def step():
    ex.clear(); ex.push_device(n, nw, ptrs, base_read_index=0); X.run(ex, n)
for _ in range(5): step()
torch.cuda.synchronize(); t = time.perf_counter()
for _ in range(100): step()
torch.cuda.synchronize(); print("step ms", (time.perf_counter() - t) * 10)
pr = cProfile.Profile(); pr.enable()
for _ in range(200): step()
torch.cuda.synchronize(); pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(22); print(s.getvalue()[:5000])
dist.destroy_process_group()
