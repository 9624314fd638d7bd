cd "$GRAFT_REPO_ROOT"
for cfg in "default:X=0" "tile4096:IPX_SORT_TILE=4096" "nt256:IPX_SCATTER_NT=256" "bsort:IPX_BSORT_MIN=1" "noxcd:IPX_SCATTER_XCD=0" "hist1:IPX_HIST_GROUP=1"; do
IFS=: read name e1 <<< "$cfg"
env $e1 python bench.py --workload S --steps 30 --warmup 5 --no-cpu-baseline --no-configs --no-parity 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('$name', round(d['ms_per_step'],4), {k:round(v*1e3) for k,v in r['kernels_ms_per_step'].items()})"
done
