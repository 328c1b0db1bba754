cd $GRAFT_REPO_ROOT
O=gpurun_out/run22; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_lowp_gpu.py -x -q -k "lp_conv_forward_stats or lazy_batchnorm or fused_upconv or first_conv" > $O/t_lp.log 2>&1; echo "lp rc=$?"; tail -4 $O/t_lp.log
grep -q passed $O/t_lp.log || exit 1
grep -q failed $O/t_lp.log && exit 1
export CTU_DT=bf16
for L in main wg16off; do
  if [ $L = main ]; then unset CTU_LIB; else export CTU_LIB=$PWD/scripts/build/lib_$L.so; fi
  for OP in "wgrad 16 16 64" "wgrad 16 16 96" "wgrad 64 16 64" "wgrad 16 16 128" "wgrad 32 32 32" "wgrad 128 32 32"; do
    echo "== $L $OP: $(timeout -k 10 120 python scripts/bench_layer.py $OP 3 30 2>&1 | tail -1)"
  done
done
unset CTU_LIB CTU_DT
for L in main wg16off main wg16off; do
  if [ $L = main ]; then unset CTUNET_HIP_LIB; else export CTUNET_HIP_LIB=$PWD/scripts/build/lib_$L.so; fi
  timeout -k 10 200 python bench.py --dtype bf16 --no-cpu-baseline --no-kernel-timer --steps 40 --warmup 5 > $O/b.json 2> $O/b.err || { echo "bench $L failed"; tail -5 $O/b.err; exit 1; }
  python -c "import json;d=json.load(open('$O/b.json'));print('bf16', '$L', round(d['ms_per_step'],4))"
done
unset CTUNET_HIP_LIB
timeout -k 10 300 python bench.py --model UNetSP --size 192 --dtype bf16 --no-cpu-baseline --no-kernel-timer --steps 10 --warmup 3 > $O/sp192.json 2> $O/sp192.err || tail -5 $O/sp192.err
python -c "import json;d=json.load(open('$O/sp192.json'));print('sp192 bf16', round(d['ms_per_step'],4))"
timeout -k 10 400 python bench.py --model UNetSP --size 256 --dtype f16 --no-cpu-baseline --no-kernel-timer --steps 5 --warmup 2 > $O/sp256.json 2> $O/sp256.err || tail -5 $O/sp256.err
python -c "import json;d=json.load(open('$O/sp256.json'));print('sp256 f16', round(d['ms_per_step'],4))"
for BT in 0 1; do
  CTUNET_BN_TAIL=$BT timeout -k 10 200 python bench.py --dtype bf16 --no-cpu-baseline --no-kernel-timer --steps 40 --warmup 5 > $O/b.json 2> $O/b.err || { echo "bench failed"; tail -5 $O/b.err; exit 1; }
  python -c "import json;d=json.load(open('$O/b.json'));print('bf16 BN_TAIL', '$BT', round(d['ms_per_step'],4))"
  CTUNET_BN_TAIL=$BT timeout -k 10 200 python bench.py --no-cpu-baseline --no-kernel-timer --steps 40 --warmup 5 > $O/b.json 2> $O/b.err || { echo "bench failed"; tail -5 $O/b.err; exit 1; }
  python -c "import json;d=json.load(open('$O/b.json'));print('f32 BN_TAIL', '$BT', round(d['ms_per_step'],4))"
done
