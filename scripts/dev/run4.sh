cd $GRAFT_REPO_ROOT
O=gpurun_out/run4; mkdir -p $O
for a in "8 8 128 1" "16 8 128 1" "16 16 64 0" "32 32 32 0"; do timeout -k 5 60 scripts/build/diag_stamp $a 2>/dev/null | tee -a $O/stamp.txt; done
timeout -k 10 600 python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "lazy" > $O/t_ops.log 2>&1; echo "ops rc=$?"; tail -3 $O/t_ops.log
timeout -k 10 900 python -m pytest tests/test_models_gpu.py -x -q -m gpu -k "not 256 and not 192" > $O/t_models.log 2>&1; echo "models rc=$?"; tail -3 $O/t_models.log
MAIN=ct-unet_amd/ctunet_amd/libctunet_hip.so
run() {  # lib, env
  env $2 CTUNET_HIP_LIB=$PWD/$1 timeout -k 10 200 python bench.py --no-cpu-baseline > $O/one.json 2>$O/one.err || { tail -5 $O/one.err; exit 1; }
  python - "$1 $2" <<'PY' | tee -a $O/bench.txt
import json,sys
d=json.loads(open('gpurun_out/run4/one.json').read().strip().splitlines()[-1])
k=d.get('kernels',{})
def g(n):
    return round(k[n]['avg_ms']*1e3,1) if n in k else None
print(sys.argv[1], 'ms/step', round(d['ms_per_step'],4), 'pair', g('conv3d_fwd_k3_persist<1, true>'), 'nt1', g('conv3d_fwd_k3_persist<1, false>'), 'wg22', g('conv3d_wgrad_k3s_kernel<2, 2> (+slab reduce)'), 'wg11', g('conv3d_wgrad_k3s_kernel<1, 1> (+slab reduce)'), 'upwg', [round(v['avg_ms']*1e3,1) for n,v in k.items() if 'upconv_fused_wgrad' in n], 'fwg', g('first_wgrad_kernel<1> (+slab reduce)'))
PY
}
for R in 1 2; do
run $MAIN CTUNET_LAZY_BN=0
run $MAIN CTUNET_LAZY_BN=1
done
