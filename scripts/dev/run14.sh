cd $GRAFT_REPO_ROOT
O=gpurun_out/run14; mkdir -p $O
for LZ in 0 1 0 1; do
  CTUNET_LAZY_BN_LP=$LZ timeout -k 10 200 python bench.py --dtype bf16 --no-cpu-baseline --no-kernel-timer --steps 40 --warmup 5 > $O/b.json 2> $O/b.err || { echo "bench failed"; tail -5 $O/b.err; exit 1; }
  python -c "import json;d=json.load(open('$O/b.json'));print('bf16 LZ', '$LZ', round(d['ms_per_step'],4))"
done
timeout -k 10 300 python scripts/stage_table.py --dtype bf16 --steps 5 --out $O/stage_bf16.md > /dev/null 2> $O/st.err || tail -5 $O/st.err
cut -c1-130 $O/stage_bf16.md | head -48
