# Round-2 measurement batch on the GPU box.  Everything lands under gpurun_out/final2/ with the names it gets in profiles/
# (scripts/collect_r02.sh copies the summaries there).  Each leg is bounded; a leg that fails stops the batch.
set -e
O=gpurun_out/final2
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
B="python bench.py"
echo "[1] default bench (headline, with cpu_baseline)"
timeout -k 10 600 $B > $O/r02_bench_default.json 2> $O/bench_default.err
echo "[2] torchrun 1 rank (segmented graph + RCCL path)"
timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $O/r02_bench_torchrun1.json 2> $O/bench_torchrun1.err
echo "[3] rocprof kernel stats of the default command"
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o r -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_prof.json 2> $O/bench_prof.err
cp $O/prof/r_kernel_stats.csv $O/r02_kernel_stats.csv
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_bf16 -o r -- python bench.py --dtype bf16 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_prof_bf16.json 2> $O/bench_prof_bf16.err
cp $O/prof_bf16/r_kernel_stats.csv $O/r02_kernel_stats_bf16.csv
echo "[4] FETCH_SIZE / WRITE_SIZE passes (eager, separate)"
timeout -k 10 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o r -- python bench.py --eager --steps 5 --warmup 2 --no-cpu-baseline > $O/pmc_fetch.json 2> $O/pmc_fetch.err
timeout -k 10 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o r -- python bench.py --eager --steps 5 --warmup 2 --no-cpu-baseline > $O/pmc_write.json 2> $O/pmc_write.err
python scripts/collect_traffic.py $O/pmc_fetch $O/pmc_write $O/r02_hbm_traffic.json > /dev/null
rm -f $O/*/*kernel_trace.csv $O/prof/*trace.csv $O/prof_bf16/*trace.csv $O/pmc_fetch/*counter_collection.csv $O/pmc_write/*counter_collection.csv
echo "[5] inference leg (cfg 2: 128^3, batch 2, eval)"
timeout -k 10 300 $B --mode infer --batch 2 --no-cpu-baseline --steps 30 --warmup 5 > $O/r02_bench_infer_b2.json 2> $O/bench_infer.err
echo "[6] 16-bit legs"
timeout -k 10 300 $B --dtype bf16 --no-cpu-baseline --steps 30 --warmup 5 > $O/r02_bench_bf16.json 2> $O/bench_bf16.err
timeout -k 10 300 $B --dtype f16 --no-cpu-baseline --steps 30 --warmup 5 > $O/r02_bench_f16.json 2> $O/bench_f16.err
echo "[7] secondary nets"
timeout -k 10 300 $B --model recAE_v2_fixed --no-cpu-baseline --steps 10 --warmup 3 > $O/r02_bench_recAE_128_f32.json 2> $O/bench_recae.err
timeout -k 10 300 $B --model UNet4_2IC --no-cpu-baseline --steps 10 --warmup 3 > $O/r02_bench_UNet4_2IC_128_f32.json 2> $O/bench_2ic.err
timeout -k 10 300 $B --model UNetSP --size 192 --dtype bf16 --no-cpu-baseline --steps 10 --warmup 3 > $O/r02_bench_UNetSP_192_bf16.json 2> $O/bench_sp192.err
timeout -k 10 300 $B --model recAE_v2_fixed --size 192 --dtype bf16 --no-cpu-baseline --steps 5 --warmup 2 > $O/r02_bench_recAE_192_bf16.json 2> $O/bench_recae192.err
timeout -k 10 300 $B --model UNetSP --size 256 --dtype f16 --no-cpu-baseline --steps 5 --warmup 2 > $O/r02_bench_UNetSP_256_f16.json 2> $O/bench_sp256.err
timeout -k 10 300 $B --model UNetSP --size 192 --no-cpu-baseline --steps 5 --warmup 2 > $O/r02_bench_UNetSP_192_f32.json 2> $O/bench_sp192f.err
timeout -k 10 300 $B --model UNetSP --size 256 --no-cpu-baseline --steps 5 --warmup 2 > $O/r02_bench_UNetSP_256_f32.json 2> $O/bench_sp256f.err
echo "[8] per-stage tables"
timeout -k 10 200 python scripts/stage_table.py --traffic $O/r02_hbm_traffic.json --out $O/r02_stage_table_f32.md > /dev/null
timeout -k 10 200 python scripts/stage_table.py --dtype bf16 --out $O/r02_stage_table_bf16.md > /dev/null
timeout -k 10 200 python scripts/stage_table.py --mode infer --batch 2 --out $O/r02_stage_table_infer_b2.md > /dev/null
timeout -k 10 200 python scripts/stage_table.py --model recAE_v2_fixed --out $O/r02_stage_table_recAE_f32.md > /dev/null
timeout -k 10 200 python scripts/stage_table.py --model recAE_v2_fixed --size 192 --dtype bf16 --steps 2 --out $O/r02_stage_table_recAE_192_bf16.md > /dev/null
echo "[9] SQ counters: fp32 roofline kernel, 16-bit conv kernels"
bash scripts/pmc_roofline_kernel.sh > $O/r02_pmc_roofline_kernel.txt 2> $O/pmc_roof.err
bash scripts/pmc_lp.sh > $O/r02_pmc_lp.txt 2> $O/pmc_lp.err
rm -rf gpurun_out/pmc?_roof gpurun_out/pmc_lp_*_[AB]
ls -la $O
