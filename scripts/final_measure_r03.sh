# Round-3 measurement batch on the GPU box.  Everything lands under gpurun_out/final3/ with the names it gets in profiles/
# (scripts/collect_r03.sh copies the summaries there).  Each leg is bounded; a leg that fails stops the batch.
# usage: bash scripts/final_measure_r03.sh [part]   part in {a, b, c}: the batch is split so each part fits one gpurun call
set -e
O=gpurun_out/final3
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
B="python bench.py"
PART=${1:-a}
pmc3() {   # pmc3 <tag> <command...>: FETCH_SIZE, WRITE_SIZE, SQ_VALU_MFMA_BUSY_CYCLES passes (separate, kernel trace only)
  T=$1; shift
  for C in FETCH_SIZE WRITE_SIZE SQ_VALU_MFMA_BUSY_CYCLES; do
    timeout -k 10 400 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $O/pmc_${T}_$C -o r -- "$@" > $O/pmc_${T}_$C.out 2> $O/pmc_${T}_$C.err
  done
}
if [ $PART = a ]; then
echo "[1] default bench (headline, with cpu_baseline)"
timeout -k 10 600 $B > $O/r03_bench_default.json 2> $O/bench_default.err
echo "[2] torchrun 1 rank (segmented graph + RCCL path)"
timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $O/r03_bench_torchrun1.json 2> $O/r03_bench_torchrun1.stderr.txt
echo "[3] rocprof kernel stats of the default command, and of the 16-bit leg"
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o r -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_prof.json 2> $O/bench_prof.err
cp $O/prof/r_kernel_stats.csv $O/r03_kernel_stats.csv
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_bf16 -o r -- python bench.py --dtype bf16 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_prof_bf16.json 2> $O/bench_prof_bf16.err
cp $O/prof_bf16/r_kernel_stats.csv $O/r03_kernel_stats_bf16.csv
echo "[4] FETCH_SIZE / WRITE_SIZE / matrix-pipe passes (eager bench.py, fp32 and bf16)"
pmc3 f32 python bench.py --eager --steps 5 --warmup 2 --no-cpu-baseline
pmc3 bf16 python bench.py --eager --dtype bf16 --steps 5 --warmup 2 --no-cpu-baseline
python scripts/collect_traffic.py $O/r03_hbm_traffic.json f32=$O/pmc_f32_FETCH_SIZE,$O/pmc_f32_WRITE_SIZE bf16=$O/pmc_bf16_FETCH_SIZE,$O/pmc_bf16_WRITE_SIZE > $O/traffic.txt
rm -rf $O/prof $O/prof_bf16 $O/pmc_f32_* $O/pmc_bf16_*
echo "[5] inference leg (cfg 2: 128^3, batch 2, eval)"
timeout -k 10 300 $B --mode infer --batch 2 --steps 30 --warmup 5 > $O/r03_bench_infer_b2.json 2> $O/bench_infer.err
echo "[6] 16-bit legs"
timeout -k 10 300 $B --dtype bf16 --cpu-check-only --steps 30 --warmup 5 > $O/r03_bench_bf16.json 2> $O/bench_bf16.err
timeout -k 10 300 $B --dtype f16 --cpu-check-only --steps 30 --warmup 5 > $O/r03_bench_f16.json 2> $O/bench_f16.err
fi
if [ $PART = b ]; then
echo "[7] secondary nets (cfg 3), cfg 4 / 5 legs"
timeout -k 10 300 $B --model recAE_v2_fixed --no-cpu-baseline --steps 10 --warmup 3 > $O/r03_bench_recAE_128_f32.json 2> $O/bench_recae.err
timeout -k 10 300 $B --model UNet4_2IC --no-cpu-baseline --steps 10 --warmup 3 > $O/r03_bench_UNet4_2IC_128_f32.json 2> $O/bench_2ic.err
timeout -k 10 500 $B --model UNetSP --size 192 --dtype bf16 --cpu-check-only --steps 10 --warmup 3 > $O/r03_bench_UNetSP_192_bf16.json 2> $O/bench_sp192.err
timeout -k 10 300 $B --model recAE_v2_fixed --size 192 --dtype bf16 --no-cpu-baseline --steps 5 --warmup 2 > $O/r03_bench_recAE_192_bf16.json 2> $O/bench_recae192.err
timeout -k 10 700 $B --model UNetSP --size 256 --dtype f16 --cpu-check-only --steps 5 --warmup 2 > $O/r03_bench_UNetSP_256_f16.json 2> $O/bench_sp256.err
timeout -k 10 300 $B --model UNetSP --size 192 --no-cpu-baseline --steps 5 --warmup 2 > $O/r03_bench_UNetSP_192_f32.json 2> $O/bench_sp192f.err
timeout -k 10 300 $B --model UNetSP --size 256 --no-cpu-baseline --steps 5 --warmup 2 > $O/r03_bench_UNetSP_256_f32.json 2> $O/bench_sp256f.err
fi
if [ $PART = c ]; then
echo "[8] per-stage tables with counter columns"
st() {   # st <name> <stage_table args...>
  N=$1; shift
  pmc3 st_$N python scripts/stage_table.py --profiled "$@"
  timeout -k 10 300 python scripts/stage_table.py "$@" --counters $O/pmc_st_${N}_FETCH_SIZE,$O/pmc_st_${N}_WRITE_SIZE,$O/pmc_st_${N}_SQ_VALU_MFMA_BUSY_CYCLES --out $O/r03_stage_table_$N.md > /dev/null 2> $O/st_$N.err
  rm -rf $O/pmc_st_${N}_*
}
st f32 --steps 3
st bf16 --dtype bf16 --steps 3
st infer_b2 --mode infer --batch 2 --steps 3
st recAE_f32 --model recAE_v2_fixed --steps 2
st recAE_192_bf16 --model recAE_v2_fixed --size 192 --dtype bf16 --steps 2
echo "[9] SQ counters: fp32 roofline kernel, 16-bit conv kernels"
bash scripts/pmc_roofline_kernel.sh > $O/r03_pmc_roofline_kernel.txt 2> $O/pmc_roof.err
bash scripts/pmc_lp.sh > $O/r03_pmc_lp.txt 2> $O/pmc_lp.err
rm -rf gpurun_out/pmc?_roof gpurun_out/pmc_lp_*_[AB]
fi
ls -la $O
