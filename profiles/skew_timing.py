"""The step under key skew: 4 M synthetic short reads over 200 000 / 1000 / 10 / 1 junction sites (Zipf-distributed
reads per site).  Run: python profiles/skew_timing.py"""
import os, sys, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import intronprospector_amd as ipa
from intronprospector_amd import synth
for n_sites in (200000, 1000, 10, 1):
    b, names, _ = synth.short_reads(4_000_000, seed=3, n_sites=n_sites)
    dev = torch.device("cuda", 0)
    keys = ("tid", "pos", "flag", "xs", "nh", "cigar_off", "n_cigar", "cigar")
    arrs = [torch.from_numpy(b[k]).to(dev) for k in keys]; ptrs = [t.data_ptr() for t in arrs]
    ex = ipa.JunctionsExtractor(targets=names, device=0)
    n, nw = len(b["tid"]), len(b["cigar"])
    ts = []
    for it in range(6):
        ex.clear(); torch.cuda.synchronize(); t0 = time.perf_counter()
        ex.push_device(n, nw, ptrs, base_read_index=0)
        rc, t = ex.finalize_raw(1)
        torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
    print("sites %7d junctions %7d step %.3f ms" % (n_sites, int(t.n_junctions), min(ts[2:])))
    ex.close()
