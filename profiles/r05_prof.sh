# rocprofv3 kernel stats of one workload: bash profiles/r05_prof.sh S [tag]
set -e
cd "$GRAFT_REPO_ROOT"
W=${1:-S}; T=${2:-prof}
O=$GRAFT_REPO_ROOT/gpurun_out/r05/${T}_$W; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rocprofv3 --kernel-trace --stats --output-format csv -d $O -- python bench.py --workload $W --steps 10 --warmup 2 --no-cpu-baseline --no-configs --no-parity > $O/run.log 2>&1
f=$(find $O -name "*kernel_stats.csv" | head -1)
cp $f gpurun_out/r05/${T}_kernel_stats_$W.csv
python - <<PY
import csv
rows=list(csv.DictReader(open("gpurun_out/r05/${T}_kernel_stats_$W.csv")))
for r in rows[:22]:
    print("%-60s calls %5s avg %9.1f us total %8.2f ms" % (r["Name"][:60], r["Calls"], float(r["AverageNs"])/1e3, float(r["TotalDurationNs"])/1e6))
PY
find $O -name "*.csv" ! -name "*kernel_stats*" -delete; rm -rf $O/*/*.db 2>/dev/null || true
