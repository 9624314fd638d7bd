# same-box A/B of several builds of libipx.so: bash profiles/r05_ab.sh WORKLOAD lib1 lib2 ...
cd "$GRAFT_REPO_ROOT"
W=$1; shift
for i in 1 2 3; do
for lib in "$@"; do
IPX_LIB_PATH=$PWD/$lib python bench.py --workload $W --steps 30 --warmup 5 --no-cpu-baseline --no-configs --no-parity 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('$lib', '$W', round(d['ms_per_step'],4), 'scan', round(r['kernels_ms_per_step']['k_scan']*1e3,1))"
done; done
