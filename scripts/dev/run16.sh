cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/run16; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_lowp_gpu.py -x -q -k "first_conv or lowp_nets" > $O/t_lp.log 2>&1; echo "lp rc=$?"; tail -4 $O/t_lp.log
timeout -k 10 200 python bench.py --dtype bf16 --no-cpu-baseline --no-kernel-timer --steps 40 --warmup 5 > $O/b.json 2> $O/b.err || { tail -5 $O/b.err; exit 1; }
python -c "import json;d=json.load(open('$O/b.json'));print('bf16 128', round(d['ms_per_step'],4))"
timeout -k 10 300 python bench.py --model UNetSP --size 192 --dtype bf16 --no-cpu-baseline --no-kernel-timer --steps 10 --warmup 3 > $O/sp192.json 2> $O/sp192.err || tail -5 $O/sp192.err
python -c "import json;d=json.load(open('$O/sp192.json'));print('sp192 bf16', round(d['ms_per_step'],4))"
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof256 -o r -- python bench.py --model UNetSP --size 256 --dtype f16 --no-cpu-baseline --no-kernel-timer --steps 5 --warmup 2 > $O/sp256.json 2> $O/sp256.err || tail -5 $O/sp256.err
python -c "import json;d=json.load(open('$O/sp256.json'));print('sp256 f16', round(d['ms_per_step'],4))"
python - <<'PY'
import csv
rows=list(csv.DictReader(open("gpurun_out/run16/prof256/r_kernel_stats.csv")))
tot=sum(float(r["TotalDurationNs"]) for r in rows)
n=10  # 2 warm eager? steps vary: print absolute per-call numbers
print("total ms", tot/1e6)
for r in rows[:40]:
    print(f"{float(r['TotalDurationNs'])/1e6:9.3f} ms  calls {int(r['Calls']):5d}  avg {float(r['AverageNs'])/1e3:9.1f} us  {r['Name'][:100]}")
PY
rm -rf $O/prof256/*trace.csv
