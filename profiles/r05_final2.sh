#!/bin/bash
# Round 5, second session, the final sources on one MI355X box: PMC traffic (FETCH_SIZE / WRITE_SIZE in separate passes) of
# S / G / L -> traffic.json stamped with these kernel sources, the driver's command with it in place, the shuffled line,
# rocprofv3 kernel stats of S / G / L / shuffled S.  Outputs under gpurun_out/r05g/.
set -e
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05g; mkdir -p $O $O/pmc
B="--no-cpu-baseline --no-configs --no-parity"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
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
cp $O/traffic.json profiles/traffic.json
python bench.py --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err; echo progress: default line with traffic
python bench.py --steps 20 --warmup 5 --shuffle --no-cpu-baseline --no-configs > $O/bench_S_shuffled.json 2>> $O/bench_default.err
for W in S G L; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$W -- python bench.py --workload $W --steps 10 --warmup 2 $B > $O/prof_$W.log 2>&1
  cp $(find $O/prof_$W -name '*kernel_stats.csv' | head -1) $O/kernel_stats_$W.csv
  echo progress: stats $W
done
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_Sshuf -- python bench.py --workload S --shuffle --steps 10 --warmup 2 $B > $O/prof_Sshuf.log 2>&1
cp $(find $O/prof_Sshuf -name '*kernel_stats.csv' | head -1) $O/kernel_stats_S_shuffled.csv
rm -rf $O/prof_? $O/prof_Sshuf
ls -la $O
# for the notebook: S's 2.3 M folded records by either sort, same box
for k in 1 2; do
  python bench.py --steps 20 --warmup 5 $B > $O/S_radix_$k.json 2>> $O/ab.err
  IPX_BSORT_MIN=1000000 python bench.py --steps 20 --warmup 5 $B > $O/S_bucket_$k.json 2>> $O/ab.err
done
python - <<'P'
import json,glob
for f in sorted(glob.glob('gpurun_out/r05g/S_*_?.json')):
    d=json.loads(open(f).read().strip().splitlines()[-1]); print(f, round(d['ms_per_step'],4))
P
