# dev (round 3): phase ablations of conv3d_fwd_k3_persist (timing-only builds, wrong results by construction)
cd $GRAFT_REPO_ROOT
O=gpurun_out/ab_abl; mkdir -p $O; : > $O/layers.txt
MAIN=ct-unet_amd/ctunet_amd/libctunet_hip.so
for R in 1 2; do
for LIB in $MAIN scripts/build/lib_noload.so scripts/build/lib_noldsw.so scripts/build/lib_nostore.so scripts/build/lib_noall3.so scripts/build/lib_nomfma.so; do
  echo "== $LIB (round $R)" >> $O/layers.txt
  for L in "8 8 128" "16 16 64"; do
    CTU_LIB=$PWD/$LIB timeout -k 10 120 python scripts/bench_layer.py fwd $L 3 30 2>/dev/null >> $O/layers.txt || exit 1
  done
done
done
cat $O/layers.txt
