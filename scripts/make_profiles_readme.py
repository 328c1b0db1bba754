"""dev: regenerate profiles/README.md from profiles/r01_bench_default.json, r01_kernel_stats.csv, r01_hbm_traffic.json.
usage: python scripts/make_profiles_readme.py <ms/step of the torchrun 1-rank run> <Mvox/s of it>"""
import csv, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
d = json.loads(open(f"{ROOT}/profiles/r01_bench_default.json").read().strip().splitlines()[-1])
rows = sorted(d["kernels"].items(), key=lambda kv: -kv[1]["ms_per_step"])
tab = "\n".join(f"| `{k}` | {v['launches_per_step']:.0f} | {v['avg_ms'] * 1e3:.1f} | {v['ms_per_step']:.3f} | {v['achieved_tflops']:.1f} | "
                f"{v['achieved_tflops'] / 157.3:.2f} |" for k, v in rows)
ks = list(csv.DictReader(open(f"{ROOT}/profiles/r01_kernel_stats.csv")))
n = [int(r["Calls"]) for r in ks if "adam_amsgrad_kernel" in r["Name"]][0]
tot = sum(float(r["TotalDurationNs"]) for r in ks) / 1e6 / n
grp = lambda pred: sum(float(r["TotalDurationNs"]) for r in ks if pred(r["Name"])) / 1e6 / n
isup = lambda s: "upconv" in s or "k3s_kernel<1, 1, 1>" in s or "k3s_kernel<1, 1, 2>" in s or "channel_sum" in s
conv = grp(lambda s: ("conv3d" in s or "first_" in s) and not isup(s))
up = grp(isup)
convt = grp(lambda s: "convt2" in s)
bn = grp(lambda s: "bn_" in s)
tr = json.load(open(f"{ROOT}/profiles/r01_hbm_traffic.json"))["kernels"]
dom = d["roofline"]["kernel"]
pk = tr[dom.split(" (")[0]]
domavg = [float(r["AverageNs"]) / 1e3 for r in ks if dom.split(" (")[0].replace(", ", ", ") in r["Name"].replace("(anonymous namespace)::", "")][0]
cb = d["cpu_baseline"]
txt = f"""# profiles/ — round 1 measurements (1× MI355X, gfx950, ROCm 7.2, fp32)

Workload: `bench.py` default — `UNet()` (1 in, 2 out, i_size 8, 4 blocks), one 128³ fp32 patch per GPU, train step =
`requires_grad_` input → forward (train-mode BN) → Dice + CE → backward → Adam(amsgrad) → grads None
(`ctunet/pytorch/Model.py:343-374`), replayed from a HIP graph.

| run | ms/step | voxels/s | note |
|---|---|---|---|
| `python bench.py` (defaults: 20 steps, 5 warm-up, HIP graph) | {d['ms_per_step']:.2f} | {d['value'] / 1e6:.1f} M | `r01_bench_default.json` (the JSON line as printed) |
| same under `python -m torch.distributed.run --nproc-per-node 1 … bench.py --gpus 1` (RCCL communicator; graph 1 → flat all-reduce → graph 2 with the fused Adam) | {float(sys.argv[1]):.2f} | {float(sys.argv[2]):.1f} M | the N>1 code path on one rank |
| CPU oracle (ATen-CPU fp32, {cb['cores']} granted cores of the GPU box, no checkpoint recompute) | {1e3 * 2097152 / cb['value']:.0f} | {cb['value'] / 1e6:.2f} M | `cpu_baseline`, kind "port"; Dice of the HIP path's hard segmentation vs the oracle's on identical weights/input: {cb['dice_vs_cpu_ref']:.7f}, max rel. output error {cb['max_rel_output_err']:.1e} |

Algorithmic work (SURVEY §8d): 271.7 GFLOP per step ⇒ **{271.7 / d['ms_per_step']:.1f} TFLOP/s whole-step = {271.7 / d['ms_per_step'] / 157.3:.2f} of the 157.3 TF
fp32-MFMA peak** — algorithmic, i.e. counted as the reference's layers; the fused decoder up-convolutions execute 3.9×
fewer multiply-adds than the two layers they replace, which is why their rows below exceed the hardware peak.

## Convolution kernels (HIP events around every launch; algorithmic FLOPs = 2·C_in·C_out·27·voxels, fused
## up-convolution = ConvTranspose + conv)

| kernel (as named in the rocprofv3 trace) | launches/step | avg µs | ms/step | algorithmic TFLOP/s | ÷ 157.3 |
|---|---|---|---|---|---|
{tab}

`conv3d_fwd_k3_persist<1, true>` (the C_out = 8 "pair" layout, the largest single symbol) executes 36 zero-padded taps
for 27: its matrix pipe runs at {d['kernels']['conv3d_fwd_k3_persist<1, true>']['achieved_tflops'] * 36 / 27:.0f} TF of MFMA work for {d['kernels']['conv3d_fwd_k3_persist<1, true>']['achieved_tflops']:.1f} TF algorithmic.

`r01_kernel_stats.csv` — `rocprofv3 --kernel-trace --stats` of `python bench.py --steps 20 --warmup 5` (all kernels;
per-step = total ÷ {n} executions, which include the 5 eagerly launched roofline steps).  GPU-busy {tot:.2f} ms/step:
plain convolutions (fwd + data-grad + weight-grad + slab reductions) {conv:.2f} ms, fused up-convolution family
(forward, dX, dW_eff, packing, projections) {up:.2f} ms, remaining ConvTranspose (two deep levels) {convt:.2f} ms,
BatchNorm finalize/backward {bn:.2f} ms, everything else ≈{tot - conv - up - convt - bn:.2f} ms.  Average duration of the roofline kernel
`{dom}` in that trace: {domavg:.1f} µs (HIP events in `bench.py`: {d['roofline']['avg_launch_ms'] * 1e3:.1f} µs).
`r01_hbm_traffic.json` — `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` (separate passes, `bench.py --eager`) per
kernel and launch, FETCH_SIZE doubled as the gfx950 guide prescribes.  The roofline kernel moves
{pk['hbm_bytes_per_launch'] / 1e6:.0f} MB per launch ({pk['read_bytes_per_launch'] / 1e6:.0f} read + {pk['write_bytes_per_launch'] / 1e6:.0f} written) against {d['roofline']['algorithmic_mb_per_launch']:.0f} MB algorithmic (halo overlap of a
4×4×32 box; FETCH_SIZE counts L2 misses that the Infinity Cache absorbs, so this is fabric traffic, not DRAM traffic);
the kernel is MFMA-bound either way.
`r01_pmc_roofline_kernel.txt` — SQ counters of the roofline kernel (`scripts/pmc_roofline_kernel.sh`): the MFMA count equals
the pair layout's 36-tap count exactly, the matrix pipe is busy 72 % of the kernel's shader cycles, VALU/MFMA co-execution is 0.
`r01_mid_kernel_stats.csv` — the same profile earlier in the round (10.1 ms/step) for comparison.

Secondary numbers (`scripts/bench_classes.py`, eagerly launched train steps at 128³, same box class): `UNet` 4.08 ms,
`UNetSP` 4.16 ms, `UNetSPSmall` (5 blocks) 4.52 ms; the legacy k = 5 nets `recAE_v2_fixed` 23.1 ms and `UNet4_2IC`
22.7 ms (≈52 algorithmic TFLOP/s: they still run the non-persistent generic kernels — next round).

History inside round 1 (ms/step): 18.8 first working path → 10.1 in-block wgrad reduction → 9.7 gather packing /
exact BN rows → 8.5 (shift, channel) wgrad tiles → 8.1 pair-layout forward → 7.7 ConvTranspose float4 stores →
7.1 HIP graph → 6.9 precomputed halo offsets + float4 conv epilogue → 6.45 fused Adam → 5.9 direct first-layer
kernels, batched weight packing → 5.5 explicit `ds_read_b64` fragment reads, hand-scheduled double buffering →
5.4 uniform interior fast paths → 5.2 pipelined persistent weight-gradient kernel for every k=3 layer →
5.05 two-stage LDS weight gradient → 4.94 BN replay folded into backward, quad-mapped head backward →
4.84 fused up-convolution forward → 4.57 + its parameter gradients → 4.09 + its data gradient and cheaper
projections → 3.98 → 3.93 w-parity-in-tile forward for 8 output channels → 3.88 the same for the weight gradient → 3.80 narrower fused
blocks where the coarse volume has fewer boxes than CUs → 3.76 parallel loss / first-layer slab reductions →
3.60 register-prefetched, ring-pipelined small-volume forward kernel → 3.56 one block per CU for small weight
gradients, shuffle finalizes, in-place loss scalars → 3.48 BatchNorm-backward reductions emitted by the max-pool and head
backward kernels → {d['ms_per_step']:.2f} batched, branch-free staging loads in the ConvTranspose kernels, output-channel quads per lane in the
composite-weight packing and projection kernels.
"""
open(f"{ROOT}/profiles/README.md", "w").write(txt)
print(txt[:600])
