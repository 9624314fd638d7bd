#!/bin/bash
# PMC traffic of one workload: two rocprofv3 runs (FETCH_SIZE / WRITE_SIZE must be separate), outputs under
# gpurun_out/pmc_<W>_{fetch,write}; then  python profiles/pmc_traffic.py W 8 gpurun_out/pmc_W_fetch gpurun_out/pmc_W_write
set -e
W=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for c in FETCH_SIZE WRITE_SIZE; do
  d=gpurun_out/pmc_${W}_$(echo $c | tr A-Z a-z | sed s/_size//)
  rm -rf $d
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $d -- python bench.py --workload $W --steps 3 --warmup 1 --no-cpu-baseline --no-configs --no-parity "$@" > $d.log 2>&1
done
