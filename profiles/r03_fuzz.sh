# Differential fuzzing of the round's final code against the oracle (profiles/fuzz_campaign.py), new seeds
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/fuzz3; mkdir -p $O
run() { name=$1; n=$2; seed=$3; shift 3; env "$@" timeout -k 10 420 python profiles/fuzz_campaign.py $n $seed > $O/$name.log 2>&1; echo "{\"run\": \"r03_fuzz:$name ($*)\", \"summary\": $(tail -n 1 $O/$name.log)}" >> $O/summary.jsonl; echo progress $name; }
rm -f $O/summary.jsonl
run default 2500 500000 IPX_NONE=1
run bsort_always 800 510000 IPX_BSORT_MIN=1
run short16_bt128 800 520000 IPX_SHORT_MAX_AVG=16 IPX_SHORT_BT=128
run short16_bt64 500 530000 IPX_SHORT_MAX_AVG=16 IPX_SHORT_BT=64
run nocombine 500 540000 IPX_COMBINE=0
run bam_records 300 550000 IPX_FUZZ_BAM=1
cat $O/summary.jsonl
