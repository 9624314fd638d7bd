cd $GRAFT_REPO_ROOT
O=gpurun_out/r03b; rm -rf $O; mkdir -p $O
python bench.py --rehearse-collectives --steps 20 --warmup 5 --no-cpu-baseline --no-configs > $O/bench_world1_exchange.json 2> $O/exch.err; echo progress exchange
python profiles/consistency.py S G L Sshuf M --rounds 40 > $O/consistency.txt 2>&1; echo progress consistency
bash profiles/cli_timeline_device_inflate.sh > $O/cli_timeline.jsonl 2> $O/cli.err; echo progress cli
python bench.py --workload M1 --steps 3 --warmup 1 > $O/bench_M1.json 2> $O/m1.err; echo progress m1
timeout -k 10 900 python -m pytest tests -q -m gpu > $O/tests.log 2>&1; tail -3 $O/tests.log
