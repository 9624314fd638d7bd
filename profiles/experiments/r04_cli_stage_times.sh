#!/bin/bash
# where the command line's time goes on the device-inflate path: the library's laps (IPX_DEBUG_PLAN) and the host's (IPX_TIMELINE)
cd "$(dirname "$0")/../.."
BIN=intronprospector_amd/bin
T=/tmp/ipx_tl2; mkdir -p $T
$BIN/ipx_host_check make-bam 10000000 $T/raw.bam > /dev/null && $BIN/ipx_host_check copy-aln $T/raw.bam $T/s.bam && rm -f $T/raw.bam
for run in 1 2 3; do
  /usr/bin/env bash -c "TIMEFORMAT='wall %R user %U sys %S'; time IPX_DEVICE_INFLATE=1 IPX_DEBUG_PLAN=1 IPX_TIMELINE=1 $BIN/intronProspector -c $T/o.tsv $T/s.bam" 2>&1 | grep -v "plan\|window"
done
rm -rf $T
