# dev: A/B the small-volume generic forward kernel across differently built libraries
for LIB in ${LIBS:-ct-unet_amd/ctunet_amd/libctunet_hip.so scripts/build/lib_unr.so}; do echo "== $LIB"; for L in "64 64 16" "32 64 16" "64 32 16" "64 128 8" "128 128 8"; do
  CTU_LIB=$PWD/$LIB python scripts/bench_layer.py fwd $L 3 50 || exit 1
done; done
