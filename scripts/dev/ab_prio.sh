# dev (round 3): A/B of the staging-priority builds on one box.  lib_p0 = no s_setprio, shipped = stage prio 1, lib_p3 = stage prio 3
cd $GRAFT_REPO_ROOT
O=gpurun_out/ab_prio; mkdir -p $O
MAIN=ct-unet_amd/ctunet_amd/libctunet_hip.so
for R in 1 2; do
for LIB in scripts/build/lib_p0.so $MAIN scripts/build/lib_p3.so; do
  echo "== $LIB (round $R)" >> $O/layers.txt
  for L in "8 8 128" "16 8 128" "16 16 64" "32 16 64" "32 32 32" "64 64 16"; do
    CTU_LIB=$PWD/$LIB timeout -k 10 120 python scripts/bench_layer.py fwd $L 3 30 >> $O/layers.txt 2>&1 || exit 1
  done
done
done
cat $O/layers.txt
for LIB in scripts/build/lib_p0.so $MAIN scripts/build/lib_p3.so scripts/build/lib_p0.so $MAIN scripts/build/lib_p3.so; do
  CTUNET_HIP_LIB=$PWD/$LIB timeout -k 10 200 python bench.py --no-cpu-baseline > $O/one.json 2>$O/one.err || { tail -5 $O/one.err; exit 1; }
  python - "$LIB" <<'PY' | tee -a $O/bench.txt
import json,sys
d=json.loads(open('gpurun_out/ab_prio/one.json').read().strip().splitlines()[-1])
k=d.get('kernels',{})
def g(n):
    return round(k[n]['avg_ms']*1e3,1) if n in k else None
print(sys.argv[1].split('/')[-1], 'ms/step', round(d['ms_per_step'],4), 'pair', g('conv3d_fwd_k3_persist<1, true>'), 'nt1', g('conv3d_fwd_k3_persist<1, false>'), 'upfwd', [round(v['avg_ms']*1e3,1) for n,v in k.items() if 'upconv_fused_fwd' in n], 'upbwd', [round(v['avg_ms']*1e3,1) for n,v in k.items() if 'upconv_fused_bwd' in n])
PY
done
