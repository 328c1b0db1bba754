cd $GRAFT_REPO_ROOT
O=gpurun_out/run27; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_lowp_gpu.py -x -q -k "conv_transpose or lowp_nets" > $O/t_lp.log 2>&1; echo "lp rc=$?"; tail -4 $O/t_lp.log
grep -q passed $O/t_lp.log || exit 1
grep -q failed $O/t_lp.log && exit 1
export CTU_DT=bf16
for OP in "convt 128 128 16" "convt_bwd 128 128 16" "convt_wgrad 128 128 16" "convt 64 64 8" "convt_bwd 64 64 8" "convt 64 64 32" "convt_bwd 64 64 32"; do
  echo "== $OP: $(timeout -k 10 120 python scripts/bench_layer.py $OP 3 30 2>&1 | tail -1)"
done
unset CTU_DT
for i in 1 2; do
timeout -k 10 200 python bench.py --dtype bf16 --no-cpu-baseline --no-kernel-timer --steps 40 --warmup 5 > $O/b.json 2> $O/b.err || { echo "bench failed"; tail -5 $O/b.err; exit 1; }
python -c "import json;d=json.load(open('$O/b.json'));print('bf16 128', round(d['ms_per_step'],4))"
done
