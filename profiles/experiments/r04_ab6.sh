cd "$GRAFT_REPO_ROOT"
for W in S G; do
for cfg in "prev:profiles/microbench/libipx_prev.so::" "ghosts_off:intronprospector_amd/libipx.so:IPX_GHOSTS=0:" "ghosts_on:intronprospector_amd/libipx.so::"; do
IFS=: read name lib e1 e2 <<< "$cfg"
env $e1 $e2 IPX_LIB_PATH=$PWD/$lib python bench.py --workload $W --steps 30 --warmup 5 --no-cpu-baseline --no-configs --no-parity 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('$W $name', round(d['ms_per_step'],4), 'recs', r.get('records_after_combine'), {k:round(v*1e3) for k,v in r['kernels_ms_per_step'].items()})"
done; done
