#!/bin/bash
# Round 5, second session, after the bucket sort took tables from 8 M records on: the full-size shuffled-S parity test,
# 40-fold consistency of S / G / L / shuffled S / the multi-push shard, a differential fuzzing run with seeds of its own
# (default switches; the bucket sort forced), the one-rank exchange rehearsal.  Outputs under gpurun_out/r05h/.
set -e
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05h; mkdir -p $O
timeout -k 10 300 python -m pytest tests/test_gpu_fullsize.py -x -q -m gpu -k "random_order" > $O/test_shuffled_10M.log 2>&1 || { tail -30 $O/test_shuffled_10M.log; exit 1; }
tail -2 $O/test_shuffled_10M.log
python profiles/consistency.py S G L Sshuf M --rounds 40 > $O/consistency.txt 2>&1; cat $O/consistency.txt | grep rounds; echo progress: consistency
python profiles/fuzz_campaign.py 800 800000 > $O/fuzz_default.log 2>&1; tail -n 1 $O/fuzz_default.log
IPX_BSORT_MIN=1 python profiles/fuzz_campaign.py 400 810000 > $O/fuzz_bsort.log 2>&1; tail -n 1 $O/fuzz_bsort.log
for f in $O/fuzz_*.log; do echo "{\"run\": \"$(basename $f .log)\", \"summary\": $(tail -n 1 $f)}"; done > $O/fuzz_summary.jsonl
python bench.py --rehearse-collectives --steps 20 --warmup 5 --no-cpu-baseline --no-configs > $O/bench_world1_exchange.json 2> $O/exch.err; echo progress: exchange
python -c "
import json; d=json.loads(open('$O/bench_world1_exchange.json').read().strip().splitlines()[-1]); print('exchange rehearsal', d['ms_per_step'])"
