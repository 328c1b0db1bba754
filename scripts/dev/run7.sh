cd $GRAFT_REPO_ROOT
O=gpurun_out/run7; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_lowp_gpu.py -x -q -m gpu -k "fused_upconv" > $O/t_up.log 2>&1; echo "upconv rc=$?"; tail -5 $O/t_up.log
for dt in bf16; do timeout -k 10 300 python bench.py --dtype $dt --no-cpu-baseline --steps 30 --warmup 5 > $O/bench_$dt.json 2>$O/bench_$dt.err || tail -5 $O/bench_$dt.err; python - $dt <<'PY'
import json,sys
d=json.loads(open(f'gpurun_out/run7/bench_{sys.argv[1]}.json').read().strip().splitlines()[-1])
print(sys.argv[1], 'ms/step', round(d['ms_per_step'],4))
for n,v in sorted(d.get('kernels',{}).items(), key=lambda kv:-kv[1]['avg_ms'])[:14]:
    print('   ', n, round(v['avg_ms']*1e3,1))
PY
done
