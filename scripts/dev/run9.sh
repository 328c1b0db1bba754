cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/run9; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_models_gpu.py -x -q -m gpu -k "mask_stable or tiny_nets" -s > $O/t_stable.log 2>&1; echo "stable rc=$?"; grep -E "worst|passed|failed|Error" $O/t_stable.log | tail
timeout -k 10 600 python -m pytest tests/test_lowp_gpu.py -x -q -m gpu -k "lazy or fused_upconv or conv_forward" > $O/t_lp.log 2>&1; echo "lp rc=$?"; tail -5 $O/t_lp.log
timeout -k 10 900 python -m pytest tests/test_lowp_gpu.py -x -q -m gpu -k "nets_against" > $O/t_nets.log 2>&1; echo "nets rc=$?"; tail -3 $O/t_nets.log
for V in 0 1 0 1; do CTUNET_LAZY_BN=$V timeout -k 10 300 python bench.py --dtype bf16 --no-cpu-baseline --steps 40 --warmup 5 > $O/b.json 2>$O/b.err || tail -5 $O/b.err; python - $V <<'PY'
import json,sys
d=json.loads(open('gpurun_out/run9/b.json').read().strip().splitlines()[-1]); print('bf16 LAZY', sys.argv[1], round(d['ms_per_step'],4))
PY
done
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_bf16 -o r -- python bench.py --dtype bf16 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_prof_bf16.json 2> $O/bench_prof_bf16.err
cp $O/prof_bf16/r_kernel_stats.csv $O/kernel_stats_bf16.csv; rm -rf $O/prof_bf16
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_f32 -o r -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_prof_f32.json 2> $O/bench_prof_f32.err
cp $O/prof_f32/r_kernel_stats.csv $O/kernel_stats_f32.csv; rm -rf $O/prof_f32
python - <<'PY'
import csv
for tag in ("bf16","f32"):
    rows=list(csv.DictReader(open(f'gpurun_out/run9/kernel_stats_{tag}.csv')))
    tot=sum(int(r['TotalDurationNs']) for r in rows)
    print(tag,'total ms/step', round(tot/35/1e6,3))
    for r in rows[:34]:
        print(f"{int(r['TotalDurationNs'])/35/1e3:8.1f} us/step  calls {int(r['Calls'])/35:5.1f}  avg {float(r['AverageNs'])/1e3:7.1f}  {r['Name'][:100]}")
PY
