#!/bin/bash
# Differential fuzzing of the round's final code against the oracle (profiles/fuzz_campaign.py: synthetic short and long
# reads, structure-free CIGARs, every parameter), several switch settings.  Outputs under gpurun_out/r05f/.
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05f; mkdir -p $O
python profiles/fuzz_campaign.py 2500 500000 > $O/fuzz_default.log 2>&1; tail -n 1 $O/fuzz_default.log
IPX_SHORT_MAX_AVG=16 python profiles/fuzz_campaign.py 1500 510000 > $O/fuzz_short16.log 2>&1; tail -n 1 $O/fuzz_short16.log
IPX_COMBINE=0 python profiles/fuzz_campaign.py 600 520000 > $O/fuzz_nocombine.log 2>&1; tail -n 1 $O/fuzz_nocombine.log
IPX_BSORT_MIN=1 python profiles/fuzz_campaign.py 800 530000 > $O/fuzz_bsort.log 2>&1; tail -n 1 $O/fuzz_bsort.log
IPX_LONG=1 python profiles/fuzz_campaign.py 400 540000 > $O/fuzz_long1.log 2>&1; tail -n 1 $O/fuzz_long1.log
IPX_SHORT_MAX_AVG=16 IPX_RANK=sort python profiles/fuzz_campaign.py 600 550000 > $O/fuzz_short16_ranksort.log 2>&1; tail -n 1 $O/fuzz_short16_ranksort.log
for f in $O/fuzz_*.log; do echo "{\"run\": \"$(basename $f .log)\", \"summary\": $(tail -n 1 $f)}"; done > $O/fuzz_summary.jsonl
cat $O/fuzz_summary.jsonl
