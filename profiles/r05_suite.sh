set -e
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05; mkdir -p $O
timeout -k 10 1100 python -m pytest tests -x -q -m gpu ${DESELECT:-} > $O/suite.log 2>&1 || { tail -40 $O/suite.log; exit 1; }
tail -5 $O/suite.log
