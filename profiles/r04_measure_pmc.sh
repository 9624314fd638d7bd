#!/bin/bash
# Round-4 measurements, part 2: PMC traffic (FETCH_SIZE / WRITE_SIZE in separate passes) of S, G and L, re-stamped into
# profiles/traffic.json's format.  Outputs under gpurun_out/r04p/.
set -e
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04p; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
cp profiles/traffic.json $O/traffic.json
for W in S G L; do
  for c in FETCH_SIZE WRITE_SIZE; do
    d=$O/pmc_${W}_$(echo $c | tr A-Z a-z | sed s/_size//)
    rocprofv3 --kernel-trace --pmc $c --output-format csv -d $d -- python bench.py --workload $W --steps 3 --warmup 1 --no-cpu-baseline --no-configs --no-parity > $d.log 2>&1
    cp $(find $d -name '*counter_collection.csv' | head -1) $O/${W}_$(echo $c | tr A-Z a-z | sed s/_size//)_counter_collection.csv
  done
  python profiles/pmc_traffic.py $W 8 $O/pmc_${W}_fetch $O/pmc_${W}_write $O/traffic.json > $O/traffic_$W.txt
  echo progress: pmc $W
done
rm -rf $O/pmc_?_fetch $O/pmc_?_write
ls -la $O
