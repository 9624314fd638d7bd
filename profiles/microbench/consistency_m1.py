"""Same input, many finalizes: the junction and bin counts of the table must not change from run to run (the number
of RECORDS may: which listed alignments share a walk pass, and with it what the combiner folds, depends on timing).
usage: consistency_m1.py gen DIR N_CHUNKS | run DIR ROUNDS   (IPX_LIB_PATH selects the library)"""
import os, sys, time
import numpy as np
import torch            # before libipx.so: the HIP runtime torch brings along must be the one the process keeps
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from intronprospector_amd import synth

def gen(d, k):
    os.makedirs(d, exist_ok=True)
    for i in range(k):
        b, names, _ = synth.short_reads(10_000_000, seed=synth.SEED_M0 + i)
        np.savez(os.path.join(d, "c%d.npz" % i), **b)
    open(os.path.join(d, "names.txt"), "w").write("\n".join(names))

def run(d, rounds):
    import torch
    import intronprospector_amd as ipa
    from intronprospector_amd import _lib as L
    from bench import upload
    dev = torch.device("cuda", 0)
    names = open(os.path.join(d, "names.txt")).read().split("\n")
    ups, sizes = [], []
    k = 0
    while os.path.exists(os.path.join(d, "c%d.npz" % k)):
        b = dict(np.load(os.path.join(d, "c%d.npz" % k)))
        ups.append(upload(b, dev)); sizes.append((len(b["tid"]), len(b["cigar"]))); k += 1
    ex = ipa.JunctionsExtractor(targets=names, device=0)
    seen = {}
    for r in range(rounds):
        ex.clear()
        base = 0
        for (na, nw), (_, ptrs) in zip(sizes, ups):
            ex.push_device(na, nw, ptrs, base_read_index=base)
            base += na
        m = ex.records_pending()
        rc, t = ex.finalize_raw(L.IPX_MEM_DEVICE)
        if rc:
            ex._check(rc)
        key = (int(t.n_junctions), int(t.n_records))
        seen[key] = seen.get(key, 0) + 1
        if r < 3 or len(seen) > 1:
            print("round", r, "junctions", key[0], "reads counted", key[1], "records after combine", int(m), flush=True)
    print(os.environ.get("IPX_LIB_PATH", "tree"), "bsort_min", os.environ.get("IPX_BSORT_MIN"), "distinct outcomes:", seen, flush=True)
    ex.close()

if __name__ == "__main__":
    if sys.argv[1] == "gen":
        gen(sys.argv[2], int(sys.argv[3]))
    else:
        run(sys.argv[2], int(sys.argv[3]))
