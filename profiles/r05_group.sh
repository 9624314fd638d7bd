set -e
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_group.py -x -q -m gpu > $O/group_tests.log 2>&1 || { tail -30 $O/group_tests.log; exit 1; }
tail -2 $O/group_tests.log
# two members on one GPU, 1 M-alignment chunks (12.5 M alignments a member), and a group of one member
python bench.py --multi group --gpus 2 --chunk 1000000 --steps 5 --warmup 2 > $O/bench_group_2members_1gpu.json 2> $O/bench_group_2.err
python bench.py --multi group --gpus 1 --chunk 1000000 --steps 5 --warmup 2 > $O/bench_group_1member.json 2> $O/bench_group_1.err
python - <<PY
import json
for f in ("bench_group_2members_1gpu","bench_group_1member"):
    d=json.loads(open("$O/%s.json"%f).read().strip().splitlines()[-1])
    print(f, round(d["ms_per_step"],3), d["config"]["uses_rccl"], d["config"]["parity"]["equals_oracle"], d["config"]["checks"]["conservation"])
PY
