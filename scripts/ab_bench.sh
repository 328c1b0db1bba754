# dev: whole-step A/B of two library builds on one box, alternating (ALT = the other build; without it the shipped library runs twice)
for LIB in ct-unet_amd/ctunet_amd/libctunet_hip.so $ALT ct-unet_amd/ctunet_amd/libctunet_hip.so $ALT; do
  CTUNET_HIP_LIB=$PWD/$LIB python bench.py --no-cpu-baseline > gpurun_out/ab_one.json 2>/dev/null || exit 1
  python -c "
import json,sys
d=json.loads(open('gpurun_out/ab_one.json').read().strip().splitlines()[-1])
k=d['kernels']
print(sys.argv[1].split('/')[-1], round(d['ms_per_step'],4), 'first_bwd', round(k['first_bwd_data_kernel<1>']['avg_ms']*1e3,1), 'pair', round(k['conv3d_fwd_k3_persist<1, true>']['avg_ms']*1e3,1), 'wg22', round(k['conv3d_wgrad_k3s_kernel<2, 2> (+slab reduce)']['avg_ms']*1e3,1))" $LIB
done
