#!/bin/bash
# SQ instruction-mix counters of one workload (two rocprofv3 --pmc runs: the sets do not fit one pass);
# then: python profiles/aggregate_sq.py gpurun_out/sq_<TAG>_1 gpurun_out/sq_<TAG>_2
# usage: run_sq.sh TAG WORKLOAD [bench.py options]
set -e
T=$1; W=$2; shift; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD --output-format csv -d gpurun_out/sq_${T}_1 -- python bench.py --workload $W --steps 3 --warmup 1 --no-cpu-baseline --no-configs --no-parity "$@" > gpurun_out/sq_${T}_1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAIT_ANY --output-format csv -d gpurun_out/sq_${T}_2 -- python bench.py --workload $W --steps 3 --warmup 1 --no-cpu-baseline --no-configs --no-parity "$@" > gpurun_out/sq_${T}_2.log 2>&1
