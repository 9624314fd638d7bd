cd "$GRAFT_REPO_ROOT"
for cfg in "g1:IPX_BUCKETS=0:IPX_GHOSTS=0" "g2:IPX_BUCKETS=0:IPX_GHOSTS=2" "g4:IPX_BUCKETS=0:IPX_GHOSTS=4" "g8:IPX_BUCKETS=0:IPX_GHOSTS=8" "g8bk:IPX_BUCKETS=1:IPX_GHOSTS=8" "g1bk:IPX_BUCKETS=2:IPX_GHOSTS=0"; do
IFS=: read name e1 e2 <<< "$cfg"
env $e1 $e2 python bench.py --workload S --steps 30 --warmup 5 --no-cpu-baseline --no-configs --no-parity 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('$name', round(d['ms_per_step'],4), 'scan', round(r['kernels_ms_per_step']['k_scan']*1e3,1), 'recs', r.get('records_after_combine'))"
done
