#!/bin/bash
# The measurements behind DESIGN.md's round-2 numbers, one MI355X box: bench lines, rocprof kernel stats, the
# exchange rehearsal, a fuzz campaign.  Outputs under gpurun_out/final/.
set -e
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/final; mkdir -p $O
python bench.py --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err
python bench.py --steps 20 --warmup 5 --shuffle --no-cpu-baseline --no-configs > $O/bench_S_shuffled.json 2>> $O/bench_default.err
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-configs --rehearse-collectives > $O/bench_S_world1_exchange.json 2>> $O/bench_default.err
IPX_COMBINE=0 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-configs --no-parity > $O/bench_S_nocombine.json 2>> $O/bench_default.err
python profiles/rehearse_exchange_timing.py > $O/exchange_stage_times_world1.jsonl 2>> $O/bench_default.err
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for W in S G L; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$W -- python bench.py --workload $W --steps 10 --warmup 2 --no-cpu-baseline --no-configs --no-parity > $O/prof_$W.log 2>&1
done
echo progress: profiles done
python profiles/fuzz_campaign.py 1200 300000 > $O/fuzz_default.log 2>&1
IPX_LONG=1 IPX_COMBINE=0 python profiles/fuzz_campaign.py 300 310000 > $O/fuzz_long1_nocombine.log 2>&1
IPX_SHORT_MAX_AVG=16 python profiles/fuzz_campaign.py 300 320000 > $O/fuzz_short16.log 2>&1
for f in $O/fuzz_*.log; do tail -n 1 $f; done
