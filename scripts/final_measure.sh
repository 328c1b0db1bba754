# round-end measurement batch on the GPU box: default bench (with cpu_baseline), 1-rank torchrun (RCCL path), rocprofv3
# kernel stats, FETCH_SIZE / WRITE_SIZE passes (eager, separate).  Outputs under gpurun_out/final/.
set -e
mkdir -p gpurun_out/final
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
echo "[1] default bench"; timeout -k 10 500 python bench.py > gpurun_out/final/bench_default.json 2> gpurun_out/final/bench_default.err
echo "[2] torchrun 1 rank"; timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/final/bench_torchrun1.json 2> gpurun_out/final/bench_torchrun1.err
echo "[3] rocprof stats"; timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/final/prof -o r -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/final/bench_prof.json 2> gpurun_out/final/bench_prof.err
echo "[4] pmc fetch"; timeout -k 10 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/final/pmc_fetch -o r -- python bench.py --eager --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/final/pmc_fetch.json 2> gpurun_out/final/pmc_fetch.err
echo "[5] pmc write"; timeout -k 10 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/final/pmc_write -o r -- python bench.py --eager --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/final/pmc_write.json 2> gpurun_out/final/pmc_write.err
rm -f gpurun_out/final/*/*kernel_trace.csv gpurun_out/final/prof/*trace.csv
ls -la gpurun_out/final gpurun_out/final/*
