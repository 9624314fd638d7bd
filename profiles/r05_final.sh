#!/bin/bash
# Round 5, second session: the state of HEAD on one MI355X box.  A: shuffled S with the radix passes against the bucket
# sort (IPX_BSORT_MIN); B: the GPU suite; C: part 1 of profiles/r05_measure.sh (bench lines, rocprofv3 kernel stats).
set -e
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05f; mkdir -p $O
B="--no-cpu-baseline --no-configs --no-parity"
for k in 1 2; do
  python bench.py --steps 20 --warmup 5 --shuffle $B > $O/shuf_lsd_$k.json 2>> $O/shuf.err
  IPX_BSORT_MIN=8000000 python bench.py --steps 20 --warmup 5 --shuffle $B > $O/shuf_bucket_$k.json 2>> $O/shuf.err
done
python - <<'P'
import json,glob
for f in sorted(glob.glob('gpurun_out/r05f/shuf_*.json')):
    d=json.loads(open(f).read().strip().splitlines()[-1]); print(f, round(d['ms_per_step'],4), {k:round(v,3) for k,v in list(d['roofline']['kernels_ms_per_step'].items())[:6]})
P
echo progress: shuffled A/B done
timeout -k 10 900 python -m pytest tests -x -q -m gpu > $O/suite.log 2>&1 || { tail -40 $O/suite.log; exit 1; }
tail -3 $O/suite.log
bash profiles/r05_measure.sh 1
cp gpurun_out/r05m/* $O/ 2>/dev/null || true
