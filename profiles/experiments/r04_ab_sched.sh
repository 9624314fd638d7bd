# k_scan_short / k_scan_long3 built with each of the AMDGPU machine scheduler's strategies (-mllvm -amdgpu-sched-strategy=...),
# same-box comparison against the default build on S and G
cd "$GRAFT_REPO_ROOT"
for W in S G; do
for i in 1 2; do
for lib in intronprospector_amd/libipx.so profiles/microbench/libipx_max-ilp.so profiles/microbench/libipx_max-memory-clause.so profiles/microbench/libipx_iterative-ilp.so profiles/microbench/libipx_iterative-minreg.so profiles/microbench/libipx_iterative-maxocc.so; do
IPX_LIB_PATH=$PWD/$lib python bench.py --workload $W --steps 30 --warmup 5 --no-cpu-baseline --no-configs 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('$lib', '$W', round(d['ms_per_step'],4), 'scan', round(r['kernels_ms_per_step']['k_scan']*1e3,1), 'parity', d['config'].get('parity_prefix',{}).get('equals_oracle'))"
done; done; done
