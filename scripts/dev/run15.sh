cd $GRAFT_REPO_ROOT
O=gpurun_out/run15; mkdir -p $O
timeout -k 10 300 python bench.py --model UNetSP --size 192 --dtype bf16 --no-cpu-baseline --steps 10 --warmup 3 > $O/sp192.json 2> $O/sp192.err || tail -5 $O/sp192.err
timeout -k 10 300 python bench.py --model UNetSP --size 256 --dtype f16 --no-cpu-baseline --steps 5 --warmup 2 > $O/sp256.json 2> $O/sp256.err || tail -5 $O/sp256.err
python - <<'PY'
import json
for f in ("sp192","sp256"):
    d=json.load(open(f"gpurun_out/run15/{f}.json"))
    print(f, round(d["ms_per_step"],3), d["roofline"]["kernel"], d["roofline"]["frac"])
    for k,v in sorted(d.get("kernels",{}).items(), key=lambda kv:-kv[1]["ms_per_step"])[:14]:
        print("   ", k, v["launches_per_step"], v["avg_ms"], v["ms_per_step"], v["achieved_gbs"])
PY
timeout -k 10 300 python scripts/stage_table.py --model UNetSP --size 192 --dtype bf16 --steps 3 --out $O/stage_sp192.md > /dev/null 2> $O/st.err || tail -5 $O/st.err
cut -c1-130 $O/stage_sp192.md | head -30
