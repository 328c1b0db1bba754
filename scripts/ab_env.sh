# dev: whole-step A/B of an environment switch on one box, alternating.  usage: VAR=CTUNET_PREPACK bash scripts/ab_env.sh
for V in 1 0 1 0; do
  env $VAR=$V python bench.py --no-cpu-baseline > gpurun_out/ab_one.json 2>/dev/null || exit 1
  python -c "
import json,sys
d=json.loads(open('gpurun_out/ab_one.json').read().strip().splitlines()[-1]); print(sys.argv[1], round(d['ms_per_step'],4))" "$VAR=$V"
done
