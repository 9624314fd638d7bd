set -e
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu ${DESELECT:-} > $O/try_parity.log 2>&1 || { tail -30 $O/try_parity.log; exit 1; }
tail -3 $O/try_parity.log
for W in ${WORKLOADS:-S G}; do
python bench.py --workload $W --steps 20 --warmup 5 --no-cpu-baseline --no-configs > $O/try_$W.json 2> $O/try_$W.err
python - <<PY
import json
d=json.loads(open("$O/try_$W.json").read().strip().splitlines()[-1]); r=d['roofline']
print("$W", round(d['ms_per_step'],4), d['config']['parity_prefix']['equals_oracle'], r['records_after_combine'], {k:round(v*1e3,1) for k,v in r['kernels_ms_per_step'].items()})
PY
done
