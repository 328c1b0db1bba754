cd $GRAFT_REPO_ROOT
O=gpurun_out/run12; mkdir -p $O
echo "== cap"; timeout -k 10 200 python scripts/dev/cap_dbg.py f16 none 2 32 1 > $O/cap.log 2>&1; echo "rc=$?"; tail -2 $O/cap.log
timeout -k 10 1100 python -m pytest tests -x -q -m gpu --durations=30 > $O/suite.log 2>&1; echo "suite rc=$?"; tail -45 $O/suite.log
