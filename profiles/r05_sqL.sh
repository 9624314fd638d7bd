#!/bin/bash
# SQ counters of the long-read scan (workload L): where k_scan_long3's wave-cycles go.  Outputs under gpurun_out/r05m/.
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05m; mkdir -p $O
B="--no-cpu-baseline --no-configs --no-parity"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rocprofv3 -L > $O/counters_avail.txt 2>&1 || true
W=L
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD --output-format csv -d $O/sq_${W}_1 -- python bench.py --workload $W --steps 3 --warmup 1 $B > $O/sq_${W}_1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAIT_ANY --output-format csv -d $O/sq_${W}_2 -- python bench.py --workload $W --steps 3 --warmup 1 $B > $O/sq_${W}_2.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_WR --output-format csv -d $O/sq_${W}_3 -- python bench.py --workload $W --steps 3 --warmup 1 $B > $O/sq_${W}_3.log 2>&1
python profiles/aggregate_sq.py $O/sq_${W}_1 $O/sq_${W}_2 $O/sq_${W}_3 > $O/sq_counters_$W.txt 2>&1 || python profiles/aggregate_sq.py $O/sq_${W}_1 $O/sq_${W}_2 > $O/sq_counters_$W.txt 2>&1 || true
grep "^k_scan_long3\|^k_bs_leaf\|^k_reduce" $O/sq_counters_$W.txt
rm -rf $O/sq_?_1 $O/sq_?_2 $O/sq_?_3
