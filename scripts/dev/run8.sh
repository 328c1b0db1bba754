cd $GRAFT_REPO_ROOT
O=gpurun_out/run8; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_lowp_gpu.py -x -q -m gpu -k "fused_upconv or conv_forward_stats or channel_maps" > $O/t_lp.log 2>&1; echo "lp rc=$?"; tail -8 $O/t_lp.log
for L in 0 1; do CTU_LAY=$L CTU_DT=bf16 timeout -k 10 120 python scripts/bench_layer.py fwd 8 8 128 3 30 2>/dev/null; done
CTU_DT=bf16 timeout -k 10 120 python scripts/bench_layer.py wgrad 8 8 128 3 30 2>/dev/null
CTU_DT=bf16 timeout -k 10 120 python scripts/bench_layer.py fwd 16 16 64 3 30 2>/dev/null
timeout -k 10 900 python -m pytest tests/test_lowp_gpu.py -x -q -m gpu -k "nets_against" > $O/t_nets.log 2>&1; echo "nets rc=$?"; tail -3 $O/t_nets.log
for dt in bf16; do timeout -k 10 300 python bench.py --dtype $dt --no-cpu-baseline --steps 30 --warmup 5 > $O/bench_$dt.json 2>$O/bench_$dt.err || tail -5 $O/bench_$dt.err; python - $dt <<'PY'
import json,sys
d=json.loads(open(f'gpurun_out/run8/bench_{sys.argv[1]}.json').read().strip().splitlines()[-1])
print(sys.argv[1], 'ms/step', round(d['ms_per_step'],4))
for n,v in sorted(d.get('kernels',{}).items(), key=lambda kv:-kv[1]['avg_ms'])[:14]:
    print('   ', n, round(v['avg_ms']*1e3,1))
PY
done
