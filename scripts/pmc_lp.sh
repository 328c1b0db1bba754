# SQ counters of the 16-bit conv kernels (forward 8 -> 8 and weight gradient 8 -> 8 at 128^3, bf16) through the C ABI
# (scripts/bench_layer.py); separate --pmc passes, kernel trace only.  Output: gpurun_out/pmc_lp_*.txt
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export CTU_DT=bf16
A="SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES"
B="SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_ANY"
for OP in "fwd 8 8 128" "wgrad 8 8 128" "fwd 32 8 128"; do
  T=$(echo $OP | tr ' ' '_')
  for P in A B; do
    eval CN=\$$P
    timeout -k 10 150 rocprofv3 --kernel-trace --pmc $CN --output-format csv -d gpurun_out/pmc_lp_${T}_$P -o r -- python scripts/bench_layer.py $OP 3 5 > gpurun_out/pmc_lp_${T}_$P.log 2>&1 || exit 1
  done
done
python - <<'PY'
import csv, glob, collections
for d in sorted(glob.glob("gpurun_out/pmc_lp_*_[AB]")):
    f = glob.glob(d + "/*counter_collection.csv") + glob.glob(d + "/*/*counter_collection.csv")
    if not f:
        continue
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
    seen = set()
    for r in csv.DictReader(open(f[0])):
        k = r["Kernel_Name"]
        if "lp_conv" not in k and "lp_wgrad" not in k:
            continue
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if (r["Dispatch_Id"], k) not in seen:
            seen.add((r["Dispatch_Id"], k)); cnt[k] += 1
    for k, v in agg.items():
        print(d, k[:70], "launches", cnt[k])
        for c, x in sorted(v.items()):
            print(f"    {c:28s} {x / cnt[k]:14.0f} per launch")
PY
