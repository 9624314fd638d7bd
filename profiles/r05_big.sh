#!/bin/bash
# Round 5, second session, last box: the bucket sort's second leaf launch takes a LIST of the large leaves
# (k_bs_big_leaves) instead of walking all leaves.  Parity first (bucket-sort tests, shuffled S and L at full size,
# consistency), then shuffled S's step; only if that is green and faster: the PMC traffic of S / G / L on these sources
# -> traffic.json, the driver's line with it, and (time permitting) kernel stats of L and shuffled S.
set -e
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05j; mkdir -p $O $O/pmc
B="--no-cpu-baseline --no-configs --no-parity"
timeout -k 10 400 python -m pytest tests/test_gpu_bsort.py tests/test_gpu_fullsize.py -x -q -m gpu -k "test_gpu_bsort or random_order or L_2M" > $O/tests.log 2>&1 || { tail -30 $O/tests.log; exit 1; }
tail -2 $O/tests.log
python profiles/consistency.py Sshuf L --rounds 40 > $O/consistency.txt 2>&1; grep rounds $O/consistency.txt
python bench.py --steps 20 --warmup 5 --shuffle $B > $O/shuf.json 2>> $O/shuf.err
python - <<'P'
import json,sys
d=json.loads(open('gpurun_out/r05j/shuf.json').read().strip().splitlines()[-1])
print('shuffled', d['ms_per_step'], {k:round(v,3) for k,v in list(d['roofline']['kernels_ms_per_step'].items())[:8]})
sys.exit(0 if d['ms_per_step'] < 1.27 else 3)
P
echo progress: validated, collecting traffic
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
timeout -k 10 100 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_Sshuf -- python bench.py --workload S --shuffle --steps 10 --warmup 2 $B > $O/prof_Sshuf.log 2>&1 && cp $(find $O/prof_Sshuf -name '*kernel_stats.csv' | head -1) $O/kernel_stats_S_shuffled.csv || true
timeout -k 10 150 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_L -- python bench.py --workload L --steps 10 --warmup 2 $B > $O/prof_L.log 2>&1 && cp $(find $O/prof_L -name '*kernel_stats.csv' | head -1) $O/kernel_stats_L.csv || true
rm -rf $O/prof_L $O/prof_Sshuf
ls $O
