#!/bin/bash
# End-to-end wall time of the command line on a 10 M-record BAM whose blocks end at record boundaries (as htslib
# writes them): the file decoded on the device (IPX_DEVICE_INFLATE=1: ipx_decode_bgzf) against the host's thread-pool
# inflate with the records parsed on the device (=0, IPX_DEVICE_PARSE=1) or on the reading thread (=0, =0).
set -e
cd "$(dirname "$0")/.."
BIN=intronprospector_amd/bin
N=${1:-10000000}
T=/tmp/ipx_tl
mkdir -p $T
$BIN/ipx_host_check make-bam $N $T/raw.bam > /dev/null
$BIN/ipx_host_check copy-aln $T/raw.bam $T/s.bam
rm -f $T/raw.bam
ls -l $T/s.bam | awk '{print "{\"bam_bytes\": " $5 "}"}'
for mode in "1 0" "0 1" "0 0"; do
  set -- $mode
  echo "{\"device_inflate\": $1, \"device_parse\": $2}"
  for run in 1 2 3; do
    /usr/bin/env bash -c "TIMEFORMAT='{\"wall_s\": %R, \"user_s\": %U, \"sys_s\": %S}'; time IPX_DEVICE_INFLATE=$1 IPX_DEVICE_PARSE=$2 IPX_TIMELINE=1 $BIN/intronProspector -c $T/out$1$2.tsv -j $T/out$1$2.bed $T/s.bam" 2>&1
  done
done
cmp $T/out10.tsv $T/out00.tsv && cmp $T/out10.bed $T/out00.bed && cmp $T/out01.tsv $T/out00.tsv && echo '{"outputs_identical": true}'
md5sum $T/out10.tsv | awk '{print "{\"calls_md5\": \"" $1 "\"}"}'
wc -l $T/out10.tsv | awk '{print "{\"rows\": " $1 "}"}'
rm -rf $T
