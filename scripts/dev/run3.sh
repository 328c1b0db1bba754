# dev (round 3): correctness of the lazy BatchNorm backward + pinned fragment reads, then A/B timings
cd $GRAFT_REPO_ROOT
O=gpurun_out/run3; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "conv3d or upconv or batchnorm or lazy" > $O/t_ops.log 2>&1; echo "ops rc=$?"; tail -3 $O/t_ops.log
timeout -k 10 600 python -m pytest tests/test_models_gpu.py -x -q -m gpu -k "not 256 and not 192" > $O/t_models.log 2>&1; echo "models rc=$?"; tail -3 $O/t_models.log
MAIN=ct-unet_amd/ctunet_amd/libctunet_hip.so
: > $O/layers.txt
for LIB in scripts/build/lib_nopin.so $MAIN scripts/build/lib_occ2.so scripts/build/lib_nopin.so $MAIN scripts/build/lib_occ2.so; do
  echo "== $LIB" >> $O/layers.txt
  for L in "8 8 128" "16 8 128" "16 16 64" "32 16 64" "32 32 32" "64 32 32"; do
    CTU_LIB=$PWD/$LIB timeout -k 10 120 python scripts/bench_layer.py fwd $L 3 30 2>/dev/null >> $O/layers.txt || exit 1
  done
done
cat $O/layers.txt
run() {  # lib, env
  env $2 CTUNET_HIP_LIB=$PWD/$1 timeout -k 10 200 python bench.py --no-cpu-baseline > $O/one.json 2>$O/one.err || { tail -5 $O/one.err; exit 1; }
  python - "$1 $2" <<'PY' | tee -a $O/bench.txt
import json,sys
d=json.loads(open('gpurun_out/run3/one.json').read().strip().splitlines()[-1])
k=d.get('kernels',{})
def g(n):
    return round(k[n]['avg_ms']*1e3,1) if n in k else None
print(sys.argv[1], 'ms/step', round(d['ms_per_step'],4), 'pair', g('conv3d_fwd_k3_persist<1, true>'), 'nt1', g('conv3d_fwd_k3_persist<1, false>'), 'wg22', g('conv3d_wgrad_k3s_kernel<2, 2> (+slab reduce)'), 'upfwd', [round(v['avg_ms']*1e3,1) for n,v in k.items() if 'upconv_fused_fwd' in n], 'upbwd', [round(v['avg_ms']*1e3,1) for n,v in k.items() if 'upconv_fused_bwd' in n], 'upwg', [round(v['avg_ms']*1e3,1) for n,v in k.items() if 'upconv_fused_wgrad' in n])
PY
}
for R in 1 2; do
run scripts/build/lib_nopin.so CTUNET_LAZY_BN=0
run $MAIN CTUNET_LAZY_BN=0
run $MAIN CTUNET_LAZY_BN=1
run scripts/build/lib_occ2.so CTUNET_LAZY_BN=1
done
