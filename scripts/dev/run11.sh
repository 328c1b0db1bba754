cd $GRAFT_REPO_ROOT
O=gpurun_out/run11; mkdir -p $O
for A in "f16 none 2 32" "f16 1073741824 2 32" "f16 none 2 32 1" "bf16 none 2 32 1" "f16 none 4 64 1" "f32 none 2 32 1"; do
  echo "== $A"; timeout -k 10 200 python scripts/dev/cap_dbg.py $A > $O/cap.log 2>&1; echo "rc=$?"; grep -v "^  File\|^$\|Extension modules" $O/cap.log | tail -6
done
