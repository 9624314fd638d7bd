"""PCIe-inclusive rate of the S workload: the batch starts in host memory (pageable numpy arrays, then
pinned tensors), ipx_push_batch copies it to the device.  Never bench.py's `value`."""
import ctypes as C, json, os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import intronprospector_amd as ipa
from intronprospector_amd import synth, _lib as L

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
batch, names, _ = synth.short_reads(n, seed=synth.SEED_S)
keys = ("tid", "pos", "flag", "xs", "nh", "cigar_off", "n_cigar", "cigar")
nbytes = sum(batch[k].nbytes for k in keys)
ex = ipa.JunctionsExtractor(targets=names, device=0)
ex.reserve(n, len(batch["cigar"]), len(batch["cigar"]) // 2 + 1024)
pinned = {k: torch.from_numpy(batch[k]).pin_memory() for k in keys}

def step(src):
    ex.clear()
    if src == "pageable":
        ex.push(batch, base_read_index=0)
    else:
        b = L.Batch(n, len(batch["cigar"]), 0, *[pinned[k].data_ptr() for k in keys], L.IPX_MEM_HOST)
        ex._check(ex.lib.ipx_push_batch(ex.h, C.byref(b)))
    rc, table = ex.finalize_raw(L.IPX_MEM_DEVICE)
    ex._check(rc)

out = {"alignments": n, "input_MB": nbytes / 1e6}
for src in ("pageable", "pinned"):
    for _ in range(2):
        step(src)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        step(src)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 5
    out[src] = {"ms_per_step": dt * 1e3, "alignments_per_s": n / dt, "input_GB_per_s": nbytes / dt / 1e9}
print(json.dumps(out))
