# dev: A/B the weight-gradient launch geometry on the small layers
for LIB in ct-unet_amd/ctunet_amd/libctunet_hip.so; do echo "== $LIB"; for L in "64 64 16" "32 64 16" "16 32 32" "32 32 32" "128 32 32"; do
  CTUNET_HIP_LIB=$PWD/$LIB python scripts/bench_layer.py wgrad $L 3 50 || exit 1
done; done
