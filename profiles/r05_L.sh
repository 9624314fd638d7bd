# L after a change: the long-read parity tests, then the bench line with its kernel classes
set -e
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05; mkdir -p $O
T=${1:-v}
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -x -q -m gpu -k "long or fuzz or random or wide" > $O/${T}_long_tests.log 2>&1 || { tail -30 $O/${T}_long_tests.log; exit 1; }
tail -2 $O/${T}_long_tests.log
python bench.py --workload L --steps 10 --warmup 3 --no-cpu-baseline --no-configs > $O/${T}_L.json 2>$O/${T}_L.err
python - <<PY
import json
d=json.loads(open("$O/${T}_L.json").read().strip().splitlines()[-1]); r=d["roofline"]
print("${T}_L", round(d["ms_per_step"],4), d["config"].get("parity_prefix"), {k:round(v*1e3,1) for k,v in r["kernels_ms_per_step"].items() if v>0.03})
PY
