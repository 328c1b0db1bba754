# dev: time the forward / data-gradient conv kernels layer by layer
for LIB in ct-unet_amd/ctunet_amd/libctunet_hip.so scripts/build/lib_head.so; do echo "== $LIB"; for L in "8 8 128" "32 8 128" "8 32 128" "16 16 64" "64 16 64" "16 64 64" "32 32 32" "128 32 32" "32 128 32" "64 64 16"; do
  CTU_LIB=$PWD/$LIB python scripts/bench_layer.py fwd $L 3 20 || exit 1
done; done
