cd "$GRAFT_REPO_ROOT"
for i in 1 2; do
for cfg in "prev:profiles/microbench/libipx_prev.so::" "new_off:intronprospector_amd/libipx.so:IPX_BUCKETS=0:" "new_bk_forced:intronprospector_amd/libipx.so:IPX_BUCKETS=2:"; do
IFS=: read name lib e1 e2 <<< "$cfg"
env $e1 $e2 IPX_LIB_PATH=$PWD/$lib python bench.py --workload S --steps 30 --warmup 5 --no-cpu-baseline --no-configs --no-parity 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('$name', round(d['ms_per_step'],4), 'scan', round(r['kernels_ms_per_step']['k_scan']*1e3,1), 'recs', r.get('records_after_combine'), {k:round(v*1e3) for k,v in r['kernels_ms_per_step'].items()})"
done; done
