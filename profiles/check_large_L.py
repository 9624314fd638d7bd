"""Bit-exactness of the long-CIGAR path at a size closer to BASELINE's L (default 1 M alignments)."""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from intronprospector_amd import synth
import intronprospector_amd as ipa
from tests import helpers
from tests.oracle_binding import Oracle
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
t0 = time.time(); batch, names, _ = synth.long_reads(n, seed=778); print("generated in %.1f s, %d CIGAR ops" % (time.time() - t0, len(batch["cigar"])), flush=True)
p = {"max_anchor_indel_size": 36}
ex = ipa.JunctionsExtractor(targets=names, **p)
cuts = np.linspace(0, n, 3).astype(int)
for a, b in zip(cuts[:-1], cuts[1:]):
    ex.push(synth.slice_batch(batch, int(a), int(b)))
t = ex.get_junctions(); ex.close(); print("device: %d junctions from %d records" % (len(t["intron_start"]), t["n_records"]), flush=True)
o = Oracle(targets=names, **p); t0 = time.time(); o.process(batch); print("oracle in %.1f s" % (time.time() - t0), flush=True)
helpers.compare_tables(t, o.table()); print("IDENTICAL at", n)
