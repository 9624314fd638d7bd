#!/bin/bash
# Second differential fuzzing run of round 5, on the code as committed at the end of the round (after the fused
# k_finish_emit_identity and the vectorised k_scan_small): new seeds, the same switch settings plus BAM pushes.
# Outputs under gpurun_out/r05f2/.
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05f2; mkdir -p $O
python profiles/fuzz_campaign.py 2500 600000 > $O/fuzz_default.log 2>&1; tail -n 1 $O/fuzz_default.log
IPX_SHORT_MAX_AVG=16 python profiles/fuzz_campaign.py 1200 610000 > $O/fuzz_short16.log 2>&1; tail -n 1 $O/fuzz_short16.log
IPX_COMBINE=0 python profiles/fuzz_campaign.py 500 620000 > $O/fuzz_nocombine.log 2>&1; tail -n 1 $O/fuzz_nocombine.log
IPX_BSORT_MIN=1 python profiles/fuzz_campaign.py 700 630000 > $O/fuzz_bsort.log 2>&1; tail -n 1 $O/fuzz_bsort.log
IPX_FUZZ_BAM=1 python profiles/fuzz_campaign.py 700 640000 > $O/fuzz_bam.log 2>&1; tail -n 1 $O/fuzz_bam.log
IPX_SHORT_MAX_AVG=16 IPX_RANK=sort python profiles/fuzz_campaign.py 500 650000 > $O/fuzz_short16_ranksort.log 2>&1; tail -n 1 $O/fuzz_short16_ranksort.log
for f in $O/fuzz_*.log; do echo "{\"run\": \"$(basename $f .log)\", \"summary\": $(tail -n 1 $f)}"; done > $O/fuzz_summary.jsonl
cat $O/fuzz_summary.jsonl
