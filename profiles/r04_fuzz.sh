# Differential fuzzing of the round's final code against the oracle (profiles/fuzz_campaign.py), new seeds
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/fuzz4; mkdir -p $O
run() { name=$1; n=$2; seed=$3; shift 3; env "$@" timeout -k 10 420 python profiles/fuzz_campaign.py $n $seed > $O/$name.log 2>&1; echo "{\"run\": \"r04_fuzz:$name ($*)\", \"summary\": $(tail -n 1 $O/$name.log)}" >> $O/summary.jsonl; echo progress $name; }
rm -f $O/summary.jsonl
run default 2500 600000 IPX_NONE=1
run bsort_always 800 610000 IPX_BSORT_MIN=1
run short16 800 620000 IPX_SHORT_MAX_AVG=16
run long1_tile4096_ranksort 500 630000 IPX_LONG=1 IPX_SORT_TILE=4096 IPX_RANK=sort
run nocombine 500 640000 IPX_COMBINE=0
run bam_records 300 650000 IPX_FUZZ_BAM=1
cat $O/summary.jsonl
