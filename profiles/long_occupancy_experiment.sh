#!/bin/bash
# NOTE (round 4): IPX_LONG_PAD was retired from libipx.so with the other experiment switches; this script reproduces the
# round-2 measurement (profiles/r02_long_scan_occupancy_vs_traffic.txt) only on a checkout of round 2 or 3.
# Is the long-read scan bound by instructions or by the fabric?  Same kernel, fewer waves per CU (extra LDS per
# block through IPX_LONG_PAD): time and FETCH_SIZE of k_scan_long* per launch.  Outputs under gpurun_out/occ_*.
# usage: long_occupancy_experiment.sh "1 0" "1 14336" "3 0" ...   (IPX_LONG value, pad bytes)
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for cfg in "$@"; do
  set -- $cfg
  d=gpurun_out/occ_v$1_pad$2
  rm -rf $d
  IPX_LONG=$1 IPX_LONG_PAD=$2 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $d -- python bench.py --workload L --steps 2 --warmup 1 --no-cpu-baseline --no-configs --no-parity > $d.log 2>&1
  echo "progress: $cfg"
done
