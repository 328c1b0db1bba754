cd $GRAFT_REPO_ROOT
O=gpurun_out/run25; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_lowp_gpu.py -x -q -k "lp_conv_forward_stats or lazy_batchnorm or first_conv" > $O/t_lp.log 2>&1; echo "lp rc=$?"; tail -4 $O/t_lp.log
grep -q passed $O/t_lp.log || exit 1
grep -q failed $O/t_lp.log && exit 1
timeout -k 10 200 python bench.py --dtype bf16 --no-cpu-baseline --no-kernel-timer --steps 40 --warmup 5 > $O/b.json 2> $O/b.err || { echo "bench failed"; tail -5 $O/b.err; exit 1; }
python -c "import json;d=json.load(open('$O/b.json'));print('bf16 128', round(d['ms_per_step'],4))"
for THR in 4000000 100000000000; do
CTUNET_FIRST_WGRAD_MFMA_MIN_VOX=$THR timeout -k 10 300 python bench.py --model UNetSP --size 192 --dtype bf16 --no-cpu-baseline --no-kernel-timer --steps 10 --warmup 3 > $O/sp192.json 2> $O/sp192.err || tail -5 $O/sp192.err
python -c "import json;d=json.load(open('$O/sp192.json'));print('sp192 bf16 thr $THR', round(d['ms_per_step'],4))"
CTUNET_FIRST_WGRAD_MFMA_MIN_VOX=$THR timeout -k 10 400 python bench.py --model UNetSP --size 256 --dtype f16 --no-cpu-baseline --no-kernel-timer --steps 5 --warmup 2 > $O/sp256.json 2> $O/sp256.err || tail -5 $O/sp256.err
python -c "import json;d=json.load(open('$O/sp256.json'));print('sp256 f16 thr $THR', round(d['ms_per_step'],4))"
done
timeout -k 10 600 python -m pytest tests/test_lowp_gpu.py -x -q -k "lowp_nets" > $O/t_nets.log 2>&1; echo "nets rc=$?"; tail -3 $O/t_nets.log
