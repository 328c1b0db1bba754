# dev: the RCCL code path on one rank, several times in a row (the capture/watchdog race is timing dependent)
for i in 1 2 3 4; do
  timeout -k 10 200 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port $((29520 + i)) bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/tr_$i.json 2> gpurun_out/tr_$i.err || { echo "run $i FAILED"; tail -5 gpurun_out/tr_$i.err; exit 1; }
  python -c "
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print('run', sys.argv[2], round(d['ms_per_step'],4), d['config']['launch'])" gpurun_out/tr_$i.json $i
done
