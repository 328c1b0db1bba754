# dev: time the wgrad kernels layer by layer (optionally A/B differently built libraries in scripts/build/lib_*.so)
for lib in ct-unet_amd/ctunet_amd/libctunet_hip.so scripts/build/lib_*.so; do
  [ -f "$lib" ] || continue
  echo "== $lib"
  for L in "32 8 128" "8 8 128" "8 16 64" "64 16 64" "16 16 64" "32 32 32" "128 32 32" "16 32 32"; do
    CTU_LIB=$PWD/$lib python scripts/bench_layer.py wgrad $L 3 10 || exit 1
  done
done
