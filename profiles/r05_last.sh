#!/bin/bash
# the tests around the bucket sort that r05_big.sh left out: config 5's share (125 M alignments, 26 M records), the exchange and
# group paths
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05k; mkdir -p $O
timeout -k 10 250 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_group.py tests/test_gpu_exchange.py -x -q -m gpu -k "config_M or test_gpu_group or test_gpu_exchange" --durations=6 > $O/tests.log 2>&1; echo rc $?
tail -12 $O/tests.log
