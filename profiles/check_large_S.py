"""Bit-exactness of the short-read path beyond the bench size: N alignments in three pushes against the oracle."""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
from intronprospector_amd import synth
import intronprospector_amd as ipa
from tests import helpers
from tests.oracle_binding import Oracle
n = int(sys.argv[1])
t0=time.time(); batch, names, _ = synth.short_reads(n, seed=777); print("gen", time.time()-t0, flush=True)
ex = ipa.JunctionsExtractor(targets=names)
cuts = np.linspace(0, n, 4).astype(int)
for a, b in zip(cuts[:-1], cuts[1:]):
    ex.push(synth.slice_batch(batch, int(a), int(b)))
t = ex.get_junctions(); ex.close(); print("device done", len(t["intron_start"]), t["n_records"], flush=True)
o = Oracle(targets=names); t0=time.time(); o.process(batch); print("oracle", time.time()-t0, flush=True)
helpers.compare_tables(t, o.table()); print("IDENTICAL at", n)
