"""Same input, many finalizes: every table must be the same, bit for bit (SHA-256 over all of its arrays).
A race inside a kernel shows here long before a parity test that runs once meets it: the divergent barrier this tool
found in k_scan_short (round 3) garbled one table in eight of the M shard and had passed every test for a round and a half.
usage (on the GPU box): python profiles/consistency.py [S G L Sshuf M] [--rounds N]"""
import argparse, os, sys, time
import numpy as np
import torch            # before libipx.so: the HIP runtime torch brings along must be the one the process keeps
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from intronprospector_amd import synth, _lib as L
import intronprospector_amd as ipa


def digest(tab):
    import hashlib
    h = hashlib.sha256()
    for k in sorted(tab):
        if isinstance(tab[k], np.ndarray):
            h.update(k.encode())
            h.update(np.ascontiguousarray(tab[k]).tobytes())
    return h.hexdigest()


def rounds_of(kind, rounds):
    dev = torch.device("cuda", 0)
    params = {}
    if kind == "M":
        parts = [synth.short_reads(10_000_000, seed=synth.SEED_M0 + i) for i in range(4)]
        names, genome = parts[0][1], None
        batches = [p[0] for p in parts]
    else:
        n = {"S": 10_000_000, "G": 10_000_000, "L": 500_000, "Sshuf": 10_000_000}[kind]
        b, names, genome = bench.make_workload("S" if kind == "Sshuf" else kind, n, 20251024)
        if kind == "Sshuf":
            b = synth.permute_batch(b, np.random.default_rng(11).permutation(len(b["tid"])))
        batches = [b]
        params = bench._oracle_params(kind if kind != "Sshuf" else "S")
    if genome is not None:
        pack = (genome.names, [(t, m, len(q)) for (t, m), q in zip(genome.twobit(), genome.seqs)])
        params.update(genome=pack, junction_filter=L.IPX_SJ_CANON)
    ups = [bench.upload(b, dev) for b in batches]
    ex = ipa.JunctionsExtractor(targets=names, device=0, **params)
    seen = {}
    t0 = time.time()
    for r in range(rounds):
        ex.clear()
        base = 0
        for b, (_, ptrs) in zip(batches, ups):
            ex.push_device(len(b["tid"]), len(b["cigar"]), ptrs, base_read_index=base)
            base += len(b["tid"])
        tab = ex.get_junctions()
        d = digest(tab)
        seen[d] = seen.get(d, 0) + 1
        if len(seen) > 1 and seen[d] == 1:
            print(kind, "round", r, "a different table:", d[:16], "junctions", len(tab["intron_start"]), flush=True)
    ex.close()
    print(kind, "%d rounds, %d distinct tables %s (%.0f s)" % (rounds, len(seen), sorted(seen.values(), reverse=True), time.time() - t0), flush=True)
    return len(seen) == 1


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("kinds", nargs="*", default=["S", "G", "L", "M"])
    ap.add_argument("--rounds", type=int, default=40)
    a = ap.parse_args()
    ok = all([rounds_of(k, a.rounds) for k in a.kinds])
    sys.exit(0 if ok else 1)
