cd $GRAFT_REPO_ROOT
O=gpurun_out/run33; mkdir -p $O
timeout -k 10 1100 python -m pytest tests -x -q -m gpu --durations=8 > $O/suite.log 2>&1; echo "suite rc=$?"; tail -14 $O/suite.log
