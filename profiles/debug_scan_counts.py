import sys
sys.path.insert(0,'/root/repo')
import numpy as np, torch
import intronprospector_amd as ipa
from intronprospector_amd import synth
batch,names,_=synth.short_reads(10_000_000)
ex=ipa.JunctionsExtractor(targets=names)
ex.push(batch)
t=ex.get_junctions()
print(len(t["intron_start"]))
