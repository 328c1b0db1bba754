cd $GRAFT_REPO_ROOT
O=gpurun_out/run31; mkdir -p $O
for RS in 0 1 0 1; do
  CTUNET_REDUCE_SIDE=$RS timeout -k 10 200 python bench.py --no-cpu-baseline --no-kernel-timer --steps 40 --warmup 5 > $O/b.json 2> $O/b.err || { echo "bench failed"; tail -8 $O/b.err; exit 1; }
  python -c "import json;d=json.load(open('$O/b.json'));print('f32 REDUCE_SIDE', '$RS', round(d['ms_per_step'],4))"
  CTUNET_REDUCE_SIDE=$RS timeout -k 10 200 python bench.py --dtype bf16 --no-cpu-baseline --no-kernel-timer --steps 40 --warmup 5 > $O/b.json 2> $O/b.err || { echo "bench failed"; tail -8 $O/b.err; exit 1; }
  python -c "import json;d=json.load(open('$O/b.json'));print('bf16 REDUCE_SIDE', '$RS', round(d['ms_per_step'],4))"
done
timeout -k 10 900 python -m pytest tests/test_models_gpu.py -x -q > $O/t_models.log 2>&1; echo "models rc=$?"; tail -3 $O/t_models.log
