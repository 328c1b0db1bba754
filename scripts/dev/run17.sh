cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/run17; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_lowp_gpu.py -x -q > $O/t_lp.log 2>&1; echo "lp rc=$?"; tail -4 $O/t_lp.log
timeout -k 10 300 python -m pytest tests/test_ops_gpu.py -x -q -k "head" > $O/t_ops.log 2>&1; echo "ops rc=$?"; tail -2 $O/t_ops.log
for L in main up4off main up4off; do
  if [ $L = main ]; then unset CTUNET_HIP_LIB; else export CTUNET_HIP_LIB=$PWD/scripts/build/lib_$L.so; fi
  timeout -k 10 200 python bench.py --dtype bf16 --no-cpu-baseline --no-kernel-timer --steps 40 --warmup 5 > $O/b.json 2> $O/b.err || { echo "bench $L failed"; tail -5 $O/b.err; exit 1; }
  python -c "import json;d=json.load(open('$O/b.json'));print('bf16', '$L', round(d['ms_per_step'],4))"
done
unset CTUNET_HIP_LIB
timeout -k 10 200 python bench.py --dtype f16 --no-cpu-baseline --no-kernel-timer --steps 40 --warmup 5 > $O/b.json 2> $O/b.err || { tail -5 $O/b.err; exit 1; }
python -c "import json;d=json.load(open('$O/b.json'));print('f16 128', round(d['ms_per_step'],4))"
timeout -k 10 300 python bench.py --model UNetSP --size 192 --dtype bf16 --no-cpu-baseline --no-kernel-timer --steps 10 --warmup 3 > $O/sp192.json 2> $O/sp192.err || tail -5 $O/sp192.err
python -c "import json;d=json.load(open('$O/sp192.json'));print('sp192 bf16', round(d['ms_per_step'],4))"
timeout -k 10 400 python bench.py --model UNetSP --size 256 --dtype f16 --no-cpu-baseline --no-kernel-timer --steps 5 --warmup 2 > $O/sp256.json 2> $O/sp256.err || tail -5 $O/sp256.err
python -c "import json;d=json.load(open('$O/sp256.json'));print('sp256 f16', round(d['ms_per_step'],4))"
for LZ in 0 1; do
CTUNET_LAZY_BN_LP=$LZ timeout -k 10 400 python bench.py --model UNetSP --size 256 --dtype f16 --no-cpu-baseline --no-kernel-timer --steps 5 --warmup 2 > $O/sp256.json 2> $O/sp256.err || tail -5 $O/sp256.err
python -c "import json;d=json.load(open('$O/sp256.json'));print('sp256 f16 LZ$LZ', round(d['ms_per_step'],4))"
done
