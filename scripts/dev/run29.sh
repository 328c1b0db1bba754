cd $GRAFT_REPO_ROOT
O=gpurun_out/run29; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_lowp_gpu.py -x -q -k "lp_conv_forward_stats" > $O/t_lp.log 2>&1; echo "lp rc=$?"; tail -4 $O/t_lp.log
grep -q passed $O/t_lp.log || exit 1
grep -q failed $O/t_lp.log && exit 1
export CTU_DT=bf16
for L in main f16off; do
  if [ $L = main ]; then unset CTU_LIB; else export CTU_LIB=$PWD/scripts/build/lib_$L.so; fi
  for OP in "fwd 16 16 64" "fwd 64 16 64" "fwd 16 64 64" "fwd 32 32 32" "fwd 16 16 128" "fwd 128 32 32"; do
    echo "== $L $OP: $(timeout -k 10 120 python scripts/bench_layer.py $OP 3 30 2>&1 | tail -1)"
  done
done
unset CTU_LIB CTU_DT
for L in main f16off main f16off; do
  if [ $L = main ]; then unset CTUNET_HIP_LIB; else export CTUNET_HIP_LIB=$PWD/scripts/build/lib_$L.so; fi
  timeout -k 10 200 python bench.py --dtype bf16 --no-cpu-baseline --no-kernel-timer --steps 40 --warmup 5 > $O/b.json 2> $O/b.err || { echo "bench $L failed"; tail -5 $O/b.err; exit 1; }
  python -c "import json;d=json.load(open('$O/b.json'));print('bf16', '$L', round(d['ms_per_step'],4))"
done
