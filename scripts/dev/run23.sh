cd $GRAFT_REPO_ROOT
O=gpurun_out/run23; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_lowp_gpu.py -x -q -k "lp_conv_forward_stats or lazy_batchnorm" > $O/t_lp.log 2>&1; echo "lp rc=$?"; tail -4 $O/t_lp.log
grep -q passed $O/t_lp.log || exit 1
grep -q failed $O/t_lp.log && exit 1
for LZ in 0 1 0 1; do
  CTUNET_LAZY_BN_LP=$LZ timeout -k 10 200 python bench.py --dtype bf16 --no-cpu-baseline --no-kernel-timer --steps 40 --warmup 5 > $O/b.json 2> $O/b.err || { echo "bench failed"; tail -5 $O/b.err; exit 1; }
  python -c "import json;d=json.load(open('$O/b.json'));print('bf16 LZ', '$LZ', round(d['ms_per_step'],4))"
done
for LZ in 0 1; do
CTUNET_LAZY_BN_LP=$LZ timeout -k 10 300 python bench.py --model UNetSP --size 192 --dtype bf16 --no-cpu-baseline --no-kernel-timer --steps 10 --warmup 3 > $O/sp192.json 2> $O/sp192.err || tail -5 $O/sp192.err
python -c "import json;d=json.load(open('$O/sp192.json'));print('sp192 bf16 LZ$LZ', round(d['ms_per_step'],4))"
CTUNET_LAZY_BN_LP=$LZ timeout -k 10 400 python bench.py --model UNetSP --size 256 --dtype f16 --no-cpu-baseline --no-kernel-timer --steps 5 --warmup 2 > $O/sp256.json 2> $O/sp256.err || tail -5 $O/sp256.err
python -c "import json;d=json.load(open('$O/sp256.json'));print('sp256 f16 LZ$LZ', round(d['ms_per_step'],4))"
done
CTUNET_LAZY_BN_LP=1 timeout -k 10 600 python -m pytest tests/test_lowp_gpu.py -x -q -k "lowp_nets" > $O/t_nets.log 2>&1; echo "nets(LZ=1) rc=$?"; tail -3 $O/t_nets.log
