#!/bin/bash
# Round-4 measurements, part 3: the exchange rehearsal, one rank's share of config 5 (M1), run-to-run consistency, the
# command line, the inflate kernel, the whole GPU suite.  Outputs under gpurun_out/r04b/.
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04b; rm -rf $O; mkdir -p $O
python bench.py --rehearse-collectives --steps 20 --warmup 5 --no-cpu-baseline --no-configs > $O/bench_world1_exchange.json 2> $O/exch.err; echo progress exchange
python profiles/rehearse_exchange_timing.py > $O/exchange_stage_times_world1.jsonl 2>> $O/exch.err; echo progress stages
python bench.py --workload M1 --steps 3 --warmup 1 > $O/bench_M1.json 2> $O/m1.err; echo progress m1
python profiles/consistency.py S G L Sshuf M --rounds 40 > $O/consistency.txt 2>&1; echo progress consistency
python profiles/gpu_inflate_timing.py > $O/gpu_inflate.json 2> $O/inflate.err; echo progress inflate
bash profiles/cli_timeline_device_inflate.sh > $O/cli_timeline.jsonl 2> $O/cli.err; echo progress cli
timeout -k 10 700 python -m pytest tests -q -m gpu > $O/tests.log 2>&1; tail -3 $O/tests.log
