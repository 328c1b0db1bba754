cd $GRAFT_REPO_ROOT
O=gpurun_out/run24; mkdir -p $O
( while true; do sleep 60; echo "alive $(date +%s)"; done ) &
HB=$!
timeout -k 10 1000 python tests/arbitrate_fullsize.py UNetSP 256 > $O/r03_arbitrate_UNetSP_256.txt 2> $O/arb.err; echo "arb rc=$?"
kill $HB
tail -3 $O/arb.err; head -40 $O/r03_arbitrate_UNetSP_256.txt
