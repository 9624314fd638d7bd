#!/bin/bash
# Round 5, second session: the bucket sort from 8 M records on (coarse ranges of ~600 sample keys).  Shuffled S by the new
# default against the old one (IPX_BSORT_MIN=16777216 -> radix passes), same box, alternating; how often a sort falls
# back (IPX_DEBUG_PLAN says so), config 5's share (its coarse ranges went from 174 to 256), then the GPU suite.
set -e
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05b; mkdir -p $O
B="--no-cpu-baseline --no-configs --no-parity"
for k in 1 2 3; do
  python bench.py --steps 20 --warmup 5 --shuffle $B > $O/shuf_new_$k.json 2>> $O/shuf.err
  [ $k = 3 ] || IPX_BSORT_MIN=16777216 python bench.py --steps 20 --warmup 5 --shuffle $B > $O/shuf_old_$k.json 2>> $O/shuf.err
done
IPX_DEBUG_PLAN=1 python bench.py --steps 80 --warmup 5 --shuffle $B > $O/shuf_new_plan.json 2> $O/shuf_plan.err
echo "sorts that fell back to the radix passes: $(grep -c 'overflowed' $O/shuf_plan.err || true) of $(grep -c 'records, key bits' $O/shuf_plan.err || true) planned"
python - <<'P'
import json,glob
for f in sorted(glob.glob('gpurun_out/r05b/shuf_*.json')):
    d=json.loads(open(f).read().strip().splitlines()[-1]); print(f, round(d['ms_per_step'],4), {k:round(v,3) for k,v in list(d['roofline']['kernels_ms_per_step'].items())[:7]})
P
echo progress: shuffled A/B done
python bench.py --workload M1 --steps 3 --warmup 1 > $O/bench_M1.json 2> $O/m1.err
python -c "
import json; d=json.loads(open('$O/bench_M1.json').read().strip().splitlines()[-1]); print('M1', d['ms_per_step'], d['roofline']['kernels_ms_per_step'])"
timeout -k 10 900 python -m pytest tests -x -q -m gpu > $O/suite.log 2>&1 || { tail -40 $O/suite.log; exit 1; }
tail -3 $O/suite.log
