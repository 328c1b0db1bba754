cd $GRAFT_REPO_ROOT
timeout -k 10 120 python scripts/dev/dbg_up4.py 8 12 32 32 ones 2>&1 | tail -18
timeout -k 10 120 python scripts/dev/dbg_up4.py 8 12 32 32 rand 2>&1 | tail -18
timeout -k 10 120 python scripts/dev/dbg_up4.py 12 12 96 32 ones 2>&1 | tail -8
