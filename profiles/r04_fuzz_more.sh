# more differential fuzzing of the round's final code (profiles/fuzz_campaign.py), seeds not used before
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/fuzz4b; mkdir -p $O
run() { name=$1; n=$2; seed=$3; shift 3; env "$@" timeout -k 10 560 python profiles/fuzz_campaign.py $n $seed > $O/$name.log 2>&1; echo "{\"run\": \"r04_fuzz_more:$name ($*)\", \"summary\": $(tail -n 1 $O/$name.log)}" >> $O/summary.jsonl; echo progress $name; }
rm -f $O/summary.jsonl
run default 5000 700000 IPX_NONE=1
run short16_nocombine 1000 710000 IPX_SHORT_MAX_AVG=16 IPX_COMBINE=0
run long1 1000 720000 IPX_LONG=1
run bam_records 600 730000 IPX_FUZZ_BAM=1
cat $O/summary.jsonl
