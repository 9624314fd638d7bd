# Same-box A/B of two builds of libipx.so (profiles/microbench/libipx_prev.so against the one in the tree),
# alternating; usage: bash profiles/ab_libs.sh [WORKLOAD]   (S, G or L; default S)
cd "$GRAFT_REPO_ROOT"
W=${1:-S}
for i in 1 2 3; do
for lib in profiles/microbench/libipx_prev.so intronprospector_amd/libipx.so; do
IPX_LIB_PATH=$PWD/$lib python bench.py --workload $W --steps 30 --warmup 5 --no-cpu-baseline --no-configs --no-parity 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('$lib', '$W', round(d['ms_per_step'],4), 'scan', round(r['kernels_ms_per_step']['k_scan']*1e3,1))"
done; done
