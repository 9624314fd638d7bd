#!/bin/bash
# Third differential fuzzing run of round 5, after the item's index word gave up its top bit (IPX_ITEM_FOLDED): new
# seeds; default switches, the bucket sort forced, the fold off, BAM pushes.  Outputs under gpurun_out/r05f3/.
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05f3; mkdir -p $O
python profiles/fuzz_campaign.py 2000 700000 > $O/fuzz_default.log 2>&1; tail -n 1 $O/fuzz_default.log
IPX_BSORT_MIN=1 python profiles/fuzz_campaign.py 600 710000 > $O/fuzz_bsort.log 2>&1; tail -n 1 $O/fuzz_bsort.log
IPX_COMBINE=0 python profiles/fuzz_campaign.py 300 720000 > $O/fuzz_nocombine.log 2>&1; tail -n 1 $O/fuzz_nocombine.log
IPX_SHORT_MAX_AVG=16 python profiles/fuzz_campaign.py 800 730000 > $O/fuzz_short16.log 2>&1; tail -n 1 $O/fuzz_short16.log
IPX_FUZZ_BAM=1 python profiles/fuzz_campaign.py 400 740000 > $O/fuzz_bam.log 2>&1; tail -n 1 $O/fuzz_bam.log
for f in $O/fuzz_*.log; do echo "{\"run\": \"$(basename $f .log)\", \"summary\": $(tail -n 1 $f)}"; done > $O/fuzz_summary.jsonl
cat $O/fuzz_summary.jsonl
