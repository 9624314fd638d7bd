set -e
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05; mkdir -p $O
for W in S G L; do
python bench.py --workload $W --steps 20 --warmup 5 --no-cpu-baseline --no-configs --no-parity > $O/base_$W.json 2> $O/base_$W.err
python - <<PY
import json
d=json.loads(open("$O/base_$W.json").read().strip().splitlines()[-1]); r=d['roofline']
print("$W", round(d['ms_per_step'],4), {k:round(v*1e3,1) for k,v in r['kernels_ms_per_step'].items()})
PY
done
python bench.py --workload S --shuffle --steps 20 --warmup 5 --no-cpu-baseline --no-configs --no-parity > $O/base_Sshuf.json 2> $O/base_Sshuf.err
python -c "
import json
d=json.loads(open('$O/base_Sshuf.json').read().strip().splitlines()[-1]); print('Sshuf', round(d['ms_per_step'],4))"
