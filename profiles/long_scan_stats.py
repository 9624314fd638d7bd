"""Turn statistics of k_scan_long3 from a build with -DIPX_LONG3_STATS (IPX_LIB_PATH=profiles/microbench/libipx_stats.so):
every 997th wave prints how many turns it took and how many lane-steps of each kind applied an op (plain steps, the
full step, the anchor-opening step), how many lanes were active and how many had no CIGAR words in their ring."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import intronprospector_amd as ipa
from intronprospector_amd import synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 200_000
batch, names, _ = synth.long_reads(n)
ex = ipa.JunctionsExtractor(targets=names, max_anchor_indel_size=36)
ex.push(batch)
t = ex.get_junctions()
print("junctions", len(t["intron_start"]), "records", t["n_records"], "ops", len(batch["cigar"]))
ex.close()
