# dev: A/B the ConvTranspose kernels on the deep-level shapes (and one large one)
for LIB in ct-unet_amd/ctunet_amd/libctunet_hip.so ${ALT:-scripts/build/lib_oldct.so}; do echo "== $LIB"; for OP in convt convt_bwd convt_wgrad; do for L in "128 128 16" "64 64 8" "32 32 32" "32 32 64"; do
  CTUNET_HIP_LIB=$PWD/$LIB python scripts/bench_layer.py $OP $L 3 50 || exit 1
done; done; done
