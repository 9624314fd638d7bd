#!/bin/bash
# The measurements behind DESIGN.md's round-5 numbers, one MI355X box.  usage: bash profiles/r05_measure.sh PART
#   1  bench lines (default with configs and CPU baseline; shuffled; M1), rocprofv3 kernel stats of S / G / L
#   2  SQ counters of S and G, PMC traffic (FETCH_SIZE / WRITE_SIZE in separate passes) of S / G / L -> traffic.json
#   3  run-to-run consistency, the exchange rehearsal, the group lines, the command line
#   4  part 2 again, then the driver's command with the fresh traffic.json in place
# Outputs under gpurun_out/r05m/.
set -e
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05m; mkdir -p $O
PART=${1:-1}
B="--no-cpu-baseline --no-configs --no-parity"
if [ "$PART" = 1 ]; then
  python bench.py --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err; echo progress: default line
  python bench.py --steps 20 --warmup 5 --shuffle --no-cpu-baseline --no-configs > $O/bench_S_shuffled.json 2>> $O/bench_default.err
  python bench.py --workload M1 --steps 3 --warmup 1 > $O/bench_M1.json 2> $O/m1.err; echo progress: m1
  cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
  for W in S G L; do
    rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$W -- python bench.py --workload $W --steps 10 --warmup 2 $B > $O/prof_$W.log 2>&1
    cp $(find $O/prof_$W -name '*kernel_stats.csv' | head -1) $O/kernel_stats_$W.csv
    echo progress: stats $W
  done
  rm -rf $O/prof_?
fi
if [ "$PART" = 2 ]; then
  cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
  for W in S G; do
    rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD --output-format csv -d $O/sq_${W}_1 -- python bench.py --workload $W --steps 3 --warmup 1 $B > $O/sq_${W}_1.log 2>&1
    rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAIT_ANY --output-format csv -d $O/sq_${W}_2 -- python bench.py --workload $W --steps 3 --warmup 1 $B > $O/sq_${W}_2.log 2>&1
    python profiles/aggregate_sq.py $O/sq_${W}_1 $O/sq_${W}_2 > $O/sq_counters_$W.txt 2>&1 || true
    echo progress: sq $W
  done
  rm -rf $O/sq_?_1 $O/sq_?_2
  mkdir -p $O/pmc
  cp profiles/traffic.json $O/traffic.json
  for W in S G L; do
    for c in FETCH_SIZE WRITE_SIZE; do
      d=$O/pmc_${W}_$(echo $c | tr A-Z a-z | sed s/_size//)
      rocprofv3 --kernel-trace --pmc $c --output-format csv -d $d -- python bench.py --workload $W --steps 3 --warmup 1 $B > $d.log 2>&1
      cp $(find $d -name '*counter_collection.csv' | head -1) $O/pmc/${W}_$(echo $c | tr A-Z a-z | sed s/_size//)_counter_collection.csv
    done
    python profiles/pmc_traffic.py $W 8 $O/pmc_${W}_fetch $O/pmc_${W}_write $O/traffic.json > $O/traffic_$W.txt
    echo progress: pmc $W
  done
  rm -rf $O/pmc_?_fetch $O/pmc_?_write
fi
if [ "$PART" = 4 ]; then
  # after a change to the kernel sources: the PMC traffic again (the stamp in traffic.json must be the sources'), then the
  # driver's command with that traffic.json in place, so that the line carries roofline.traffic
  bash profiles/r05_measure.sh 2
  cp $O/traffic.json profiles/traffic.json
  python bench.py --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err; echo progress: default line with traffic
fi
if [ "$PART" = 3 ]; then
  python profiles/consistency.py S G L Sshuf M --rounds 40 > $O/consistency.txt 2>&1; echo progress: consistency
  python bench.py --rehearse-collectives --steps 20 --warmup 5 --no-cpu-baseline --no-configs > $O/bench_world1_exchange.json 2> $O/exch.err; echo progress: exchange
  python profiles/rehearse_exchange_timing.py > $O/exchange_stage_times_world1.jsonl 2>> $O/exch.err || true
  python bench.py --multi group --gpus 2 --chunk 1000000 --steps 5 --warmup 2 > $O/bench_group_2members_1gpu.json 2> $O/group.err
  python bench.py --multi group --gpus 1 --chunk 1000000 --steps 5 --warmup 2 > $O/bench_group_1member.json 2>> $O/group.err; echo progress: group
  bash profiles/cli_timeline_device_inflate.sh > $O/cli_timeline.jsonl 2> $O/cli.err || true; echo progress: cli
fi
ls -la $O
