#!/bin/bash
# Round 5, second session, final sources: the exchange path's stage times with one rank, config 5 through ipx_group (two
# members sharing the GPU; one member).  Outputs under gpurun_out/r05i/.
set -e
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05i; mkdir -p $O
python profiles/rehearse_exchange_timing.py > $O/exchange_stage_times_world1.jsonl 2> $O/exch.err || true; echo progress: stages
python bench.py --multi group --gpus 2 --chunk 1000000 --steps 5 --warmup 2 > $O/bench_group_2members_1gpu.json 2> $O/group.err
python bench.py --multi group --gpus 1 --chunk 1000000 --steps 5 --warmup 2 > $O/bench_group_1member.json 2>> $O/group.err; echo progress: group
python - <<'P'
import json
for f in ('bench_group_2members_1gpu','bench_group_1member'):
    d=json.loads(open('gpurun_out/r05i/%s.json'%f).read().strip().splitlines()[-1]); print(f, d['ms_per_step'], d.get('checks'))
P
tail -3 $O/exchange_stage_times_world1.jsonl | cut -c1-400
