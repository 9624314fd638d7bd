"""Stage times of the multi-GPU record exchange (distributed.RecordExchanger.run) with a one-rank RCCL group on
one GPU: everything but the link traffic.  Run: python profiles/rehearse_exchange_timing.py"""
import os, sys, time, json
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29544")
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
import intronprospector_amd as ipa
from intronprospector_amd import synth, distributed
batch, names, _ = synth.short_reads(10_000_000, seed=synth.SEED_S)
keys = ("tid", "pos", "flag", "xs", "nh", "cigar_off", "n_cigar", "cigar")
arrs = [torch.from_numpy(batch[k]).to(dev) for k in keys]; ptrs = [t.data_ptr() for t in arrs]
n = len(batch["tid"]); nw = len(batch["cigar"])
ex = ipa.JunctionsExtractor(targets=names, device=0); ex.reserve(0, 0, nw // 2 + 1024)
for reuse, slotted in ((True, True), (True, False), (False, False)):
    X = distributed.RecordExchanger(dev, comm_device=dev, reuse_splitters=reuse)
    X.slotted = slotted
    X.strict = False
    def step():
        t = time.perf_counter()
        ex.clear(); ex.push_device(n, nw, ptrs, base_read_index=0)
        torch.cuda.synchronize()
        push = (time.perf_counter() - t) * 1e3
        X.run(ex, n)
        return push
    for _ in range(3): step()
    X.timed = True; X.stage_ms = {}
    push = sum(step() for _ in range(10)) / 10
    out = {"clear+push": round(push, 3)}
    out.update({k: round(v / 10, 3) for k, v in X.stage_ms.items()})
    # untimed form: what a step costs when nothing synchronises between the stages
    X.timed = False
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(20):
        ex.clear(); ex.push_device(n, nw, ptrs, base_read_index=0); X.run(ex, n)
    torch.cuda.synchronize()
    print(json.dumps({"reuse_splitters": reuse, "slotted": slotted, "stages_ms": out, "total_ms": round(sum(out.values()), 3),
                      "step_ms_untimed": round((time.perf_counter() - t) / 20 * 1e3, 3)}))
dist.destroy_process_group()
