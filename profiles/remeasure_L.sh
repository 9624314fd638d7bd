set -e
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/final; mkdir -p $O
python bench.py --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_L -- python bench.py --workload L --steps 10 --warmup 2 --no-cpu-baseline --no-configs --no-parity > $O/prof_L.log 2>&1
bash profiles/run_pmc.sh L
echo remeasure done
