# a longer differential fuzz of the round's final code (profiles/fuzz_campaign.py), further seeds, in chunks that report
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/fuzz4; mkdir -p $O
run() { name=$1; n=$2; seed=$3; shift 3; t0=$(date +%s); env "$@" timeout -k 10 300 python profiles/fuzz_campaign.py $n $seed > $O/$name.log 2>&1; echo "{\"run\": \"r03_fuzz_more:$name ($*)\", \"summary\": $(tail -n 1 $O/$name.log)}" >> $O/summary.jsonl; echo progress $name $(( $(date +%s) - t0 )) s; }
rm -f $O/summary.jsonl
run default_a 1500 600000 IPX_NONE=1
run default_b 1500 602000 IPX_NONE=1
run default_c 1500 604000 IPX_NONE=1
run bsort_always 1500 620000 IPX_BSORT_MIN=1
run short16_bt128 1500 630000 IPX_SHORT_MAX_AVG=16 IPX_SHORT_BT=128
cat $O/summary.jsonl
