import sys, os
sys.path.insert(0, os.getcwd())
import intronprospector_amd as ipa
from intronprospector_amd import synth
from tests import helpers
from tests.oracle_binding import Oracle
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
batch, names, _ = synth.short_reads(n, seed=7, n_targets=3, target_len=4_000_000, n_sites=600)
ex = ipa.JunctionsExtractor(targets=names)
ex.push(batch)
dev = ex.get_junctions()
ex.close()
ora = Oracle(targets=names)
ora.process(batch)
tab = ora.table()
ora.close()
helpers.compare_tables(dev, tab)
print("ok", dev["n_alignments"], dev["n_records"], len(dev["intron_start"]))
