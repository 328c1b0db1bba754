cd $GRAFT_REPO_ROOT
export CTU_DT=bf16
for L in f16v1 f16v2; do
  export CTU_LIB=$PWD/scripts/build/lib_$L.so
  for OP in "fwd 16 16 64" "fwd 64 16 64" "fwd 16 16 128"; do
    echo "== $L $OP: $(timeout -k 10 120 python scripts/bench_layer.py $OP 3 30 2>&1 | tail -1)"
  done
done
