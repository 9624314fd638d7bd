"""Which scan kernel for batches between the two regimes?  Random CIGARs of lo..hi ops (tests/helpers.fuzz_batch),
400 k alignments tiled to 2 M; prints the scan time (ipx_last_timing push) with the short-CIGAR kernel (+ second
stage for alignments of more than 8 ops) and with the long-CIGAR kernel (IPX_SHORT_MAX_AVG=1 routes everything
there).  Run: python profiles/mid_cigar_timing.py"""
import os, subprocess, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

def child(lo, hi):
    import numpy as np, torch
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
    import helpers
    import intronprospector_amd as ipa
    from intronprospector_amd import synth
    b = helpers.fuzz_batch(400_000, seed=7, min_ops=lo, max_ops=hi)
    batch = synth.concat_batches([b] * 5)
    ex = ipa.JunctionsExtractor(targets=["chrB", "chrA"], device=0, min_anchor_length=3, min_intron_length=70)
    best = 1e9
    for _ in range(4):
        ex.clear()
        ex.push(batch)
        best = min(best, ex.timing()[0])
    t = ex.get_junctions()
    print(json.dumps({"ops": [lo, hi], "words_per_alignment": len(batch["cigar"]) / len(batch["tid"]),
                      "route": os.environ.get("IPX_SHORT_MAX_AVG", "16"), "push_ms": best,
                      "junctions": int(len(t["intron_start"]))}))

if __name__ == "__main__":
    if len(sys.argv) == 3:
        child(int(sys.argv[1]), int(sys.argv[2]))
    else:
        for lo, hi in ((2, 8), (4, 12), (6, 16), (10, 20)):
            for route in ("16", "1"):
                env = dict(os.environ, IPX_SHORT_MAX_AVG=route)
                subprocess.run([sys.executable, __file__, str(lo), str(hi)], env=env, check=True)
