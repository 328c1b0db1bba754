cd $GRAFT_REPO_ROOT
O=gpurun_out/run32; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_lowp_gpu.py -x -q -k "conv_transpose" > $O/t_lp.log 2>&1; echo "lp rc=$?"; tail -3 $O/t_lp.log
grep -q passed $O/t_lp.log || exit 1
grep -q failed $O/t_lp.log && exit 1
export CTU_DT=bf16
for L in main ctw0; do
  if [ $L = main ]; then unset CTU_LIB; else export CTU_LIB=$PWD/scripts/build/lib_$L.so; fi
  for OP in "convt_wgrad 64 64 32" "convt_wgrad 128 128 16" "convt_wgrad 64 64 8" "convt_wgrad 64 64 48"; do
    echo "== $L $OP: $(timeout -k 10 120 python scripts/bench_layer.py $OP 3 30 2>&1 | tail -1)"
  done
done
unset CTU_LIB CTU_DT
for L in main ctw0 main ctw0; do
  if [ $L = main ]; then unset CTUNET_HIP_LIB; else export CTUNET_HIP_LIB=$PWD/scripts/build/lib_$L.so; fi
  timeout -k 10 200 python bench.py --dtype bf16 --no-cpu-baseline --no-kernel-timer --steps 40 --warmup 5 > $O/b.json 2> $O/b.err || { echo "bench $L failed"; tail -5 $O/b.err; exit 1; }
  python -c "import json;d=json.load(open('$O/b.json'));print('bf16', '$L', round(d['ms_per_step'],4))"
done
