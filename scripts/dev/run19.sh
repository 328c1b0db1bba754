cd $GRAFT_REPO_ROOT
O=gpurun_out/run19; mkdir -p $O
timeout -k 10 120 python scripts/dev/dbg_up4.py 8 12 32 32 rand 2>&1 | tail -8
timeout -k 10 900 python -m pytest tests/test_lowp_gpu.py -x -q -k "fused_upconv or lowp_nets or overflow or loss_scale" > $O/t_lp.log 2>&1; echo "lp rc=$?"; tail -4 $O/t_lp.log
