cd $GRAFT_REPO_ROOT
O=gpurun_out/run35; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_ops_gpu.py tests/test_models_gpu.py -x -q -k "loss or tiny or graph or handler or ini" > $O/t.log 2>&1; echo "rc=$?"; tail -3 $O/t.log
for i in 1 2; do
timeout -k 10 200 python bench.py --no-cpu-baseline --steps 40 --warmup 5 > $O/b.json 2> $O/b.err || { tail -5 $O/b.err; exit 1; }
python -c "import json;d=json.load(open('$O/b.json'));print('f32', round(d['ms_per_step'],4))"
timeout -k 10 200 python bench.py --dtype bf16 --no-cpu-baseline --no-kernel-timer --steps 40 --warmup 5 > $O/b.json 2> $O/b.err || { tail -5 $O/b.err; exit 1; }
python -c "import json;d=json.load(open('$O/b.json'));print('bf16', round(d['ms_per_step'],4))"
done
timeout -k 10 400 python bench.py --model UNetSP --size 256 --dtype f16 --no-cpu-baseline --no-kernel-timer --steps 5 --warmup 2 > $O/sp256.json 2> $O/sp256.err || tail -5 $O/sp256.err
python -c "import json;d=json.load(open('$O/sp256.json'));print('sp256 f16', round(d['ms_per_step'],4))"
timeout -k 10 300 python bench.py --model UNetSP --size 192 --dtype bf16 --no-cpu-baseline --no-kernel-timer --steps 10 --warmup 3 > $O/sp192.json 2> $O/sp192.err || tail -5 $O/sp192.err
python -c "import json;d=json.load(open('$O/sp192.json'));print('sp192 bf16', round(d['ms_per_step'],4))"
