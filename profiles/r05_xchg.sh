set -e
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_exchange.py tests/test_gpu_shards.py -x -q -m gpu > $O/xchg_tests.log 2>&1 || { tail -40 $O/xchg_tests.log; exit 1; }
tail -2 $O/xchg_tests.log
python bench.py --rehearse-collectives --steps 20 --warmup 5 --no-cpu-baseline --no-configs > $O/xchg_world1.json 2> $O/xchg.err
python profiles/rehearse_exchange_timing.py > $O/xchg_stages.jsonl 2>> $O/xchg.err || true
python -c "
import json
d=json.loads(open('$O/xchg_world1.json').read().strip().splitlines()[-1]); print('world1', d['ms_per_step'])"
head -1 $O/xchg_stages.jsonl
