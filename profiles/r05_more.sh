set -e
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05; mkdir -p $O
T=${1:-v}
python bench.py --workload S --shuffle --steps 20 --warmup 5 --no-cpu-baseline --no-configs > $O/${T}_Sshuf.json 2>$O/${T}_Sshuf.err
python bench.py --workload M1 --steps 5 --warmup 2 --no-cpu-baseline --no-configs > $O/${T}_M1.json 2>$O/${T}_M1.err
python bench.py --workload L --steps 10 --warmup 3 --no-cpu-baseline --no-configs > $O/${T}_L.json 2>$O/${T}_L.err
python - <<PY
import json
for f in ("${T}_Sshuf","${T}_M1","${T}_L"):
    d=json.loads(open("$O/%s.json"%f).read().strip().splitlines()[-1]); r=d["roofline"]
    print(f, round(d["ms_per_step"],4), d["config"].get("parity_prefix"), r.get("records_after_combine"), {k:round(v*1e3,1) for k,v in r["kernels_ms_per_step"].items() if v>0.03})
PY
