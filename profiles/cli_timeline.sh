#!/bin/bash
# End-to-end wall time of the command line on a 10 M-record BAM (coordinate-sorted synthetic short reads),
# with the host's own phase sums (IPX_TIMELINE=1): profiles/r02_cli_timeline.jsonl
#   read_s       time in AlnReader::read on the parsing thread (BGZF inflate runs on a thread pool beside it)
#   queue_s      process_alignment: the record into the pinned SoA batch; with the block-parallel feeder (BAM input,
#                no pass-through / trace): the time the loop waited for the parse workers before a push
#                (read_s is then the splitting thread: record lengths, waiting for inflated blocks)
#   push_wait_s  time the loop spent inside ipx_push_batch_async (completing the previous batch, queuing this one)
#   finalize_s   ipx_finalize calls (sort + reduce + table to pinned host memory)
set -e
cd "$(dirname "$0")/.."
BIN=intronprospector_amd/bin
N=${1:-10000000}
T=/tmp/ipx_tl
mkdir -p $T
$BIN/ipx_host_check make-bam $N $T/s.bam > /dev/null
ls -l $T/s.bam | awk '{print "{\"bam_bytes\": " $5 "}"}'
# IPX_DEVICE_PARSE=1: the records are decoded on the device (ipx_push_bam); 0: parsed on the reading thread
for mode in 1 0; do
  echo "{\"device_parse\": $mode}"
  for run in 1 2 3; do
    /usr/bin/env bash -c "TIMEFORMAT='{\"wall_s\": %R, \"user_s\": %U, \"sys_s\": %S}'; time IPX_DEVICE_PARSE=$mode IPX_TIMELINE=1 $BIN/intronProspector -c $T/out$mode.tsv -j $T/out$mode.bed $T/s.bam" 2>&1
  done
done
cmp $T/out1.tsv $T/out0.tsv && cmp $T/out1.bed $T/out0.bed && echo '{"outputs_identical": true}'
cp $T/out1.tsv $T/out.tsv
# the BGZF/BAM decode alone, against inflate threads, all on this one file (the r01 series mixed two files)
for t in 1 2 4 8 12 16; do $BIN/ipx_host_check bench-decode $T/s.bam $t; done
md5sum $T/out.tsv | awk '{print "{\"calls_md5\": \"" $1 "\"}"}'
wc -l $T/out.tsv | awk '{print "{\"rows\": " $1 "}"}'
rm -rf $T
