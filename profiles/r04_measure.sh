#!/bin/bash
# The measurements behind DESIGN.md's round-4 numbers, one MI355X box, part 1: bench lines, rocprofv3 kernel stats, SQ
# counters of S and G.  Outputs under gpurun_out/r04/.
set -e
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04; rm -rf $O; mkdir -p $O
python bench.py --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err
python bench.py --steps 20 --warmup 5 --shuffle --no-cpu-baseline --no-configs > $O/bench_S_shuffled.json 2>> $O/bench_default.err
echo progress: bench lines done
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for W in S G L; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$W -- python bench.py --workload $W --steps 10 --warmup 2 --no-cpu-baseline --no-configs --no-parity > $O/prof_$W.log 2>&1
  cp $(find $O/prof_$W -name '*kernel_stats.csv' | head -1) $O/kernel_stats_$W.csv
  echo progress: stats $W
done
for W in S G; do
  rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD --output-format csv -d $O/sq_${W}_1 -- python bench.py --workload $W --steps 3 --warmup 1 --no-cpu-baseline --no-configs --no-parity > $O/sq_${W}_1.log 2>&1
  rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAIT_ANY --output-format csv -d $O/sq_${W}_2 -- python bench.py --workload $W --steps 3 --warmup 1 --no-cpu-baseline --no-configs --no-parity > $O/sq_${W}_2.log 2>&1
  python profiles/aggregate_sq.py $O/sq_${W}_1 $O/sq_${W}_2 > $O/sq_counters_$W.txt 2>&1 || true
  echo progress: sq $W
done
rm -rf $O/prof_? $O/sq_?_1 $O/sq_?_2
ls -la $O
