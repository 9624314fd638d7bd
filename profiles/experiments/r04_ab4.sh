cd "$GRAFT_REPO_ROOT"
for W in L G; do
for cfg in "off:IPX_BUCKETS=0" "on:IPX_BUCKETS=1"; do
IFS=: read name e1 <<< "$cfg"
env $e1 IPX_DEBUG_PLAN=1 python bench.py --workload $W --steps 10 --warmup 3 --no-cpu-baseline --no-configs --no-parity 2> gpurun_out/r4_ab4_${W}_$name.err | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('$W $name', round(d['ms_per_step'],4), {k:round(v*1e3) for k,v in r['kernels_ms_per_step'].items()})"
grep "position\|beyond" gpurun_out/r4_ab4_${W}_$name.err | sort | uniq -c | head -3
done; done
