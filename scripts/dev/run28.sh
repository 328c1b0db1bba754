cd $GRAFT_REPO_ROOT
O=gpurun_out/run28; mkdir -p $O
( while true; do sleep 60; echo "alive"; done ) &
HB=$!
CTUNET_FULLSIZE_ALL=1 timeout -k 10 1000 python -m pytest tests/test_full_size_gpu.py -x -q -k "UNet4_2IC or UNetSPSmall" --durations=4 > $O/all.log 2>&1; echo "rc=$?"
kill $HB
tail -8 $O/all.log
