"""dev: regenerate the "Round 3" section of profiles/README.md from the r03_* files (everything above "## Round 2").
usage: python scripts/make_profiles_readme_r03.py"""
import csv
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
R = ROOT + "/profiles/"


def J(n):
    return json.loads(open(R + n).read().strip().splitlines()[-1])


d = J("r03_bench_default.json"); t1 = J("r03_bench_torchrun1.json"); inf = J("r03_bench_infer_b2.json")
bf = J("r03_bench_bf16.json"); f16 = J("r03_bench_f16.json"); rec = J("r03_bench_recAE_128_f32.json")
ic = J("r03_bench_UNet4_2IC_128_f32.json"); sp192 = J("r03_bench_UNetSP_192_bf16.json")
rec192 = J("r03_bench_recAE_192_bf16.json"); sp256 = J("r03_bench_UNetSP_256_f16.json")
sp192f = J("r03_bench_UNetSP_192_f32.json"); sp256f = J("r03_bench_UNetSP_256_f32.json")
cb = d["cpu_baseline"]


def stats(name):
    ks = list(csv.DictReader(open(R + name)))
    n = [int(r["Calls"]) for r in ks if "adam_amsgrad" in r["Name"]][0]
    tot = sum(float(r["TotalDurationNs"]) for r in ks) / 1e6 / n
    grp = lambda pred: sum(float(r["TotalDurationNs"]) for r in ks if pred(r["Name"])) / 1e6 / n
    return ks, n, tot, grp


ks, n, tot, grp = stats("r03_kernel_stats.csv")
isup = lambda s: "upconv" in s or "k3s_kernel<1, 1, 1" in s or "k3s_kernel<1, 1, 2" in s
g_conv = grp(lambda s: ("conv3d" in s or "first_" in s) and not isup(s)); g_up = grp(isup); g_ct = grp(lambda s: "convt2" in s)
g_app = grp(lambda s: "bwd_apply" in s); g_red = grp(lambda s: "bwd_reduce" in s); g_fin = grp(lambda s: "finalize" in s)
g_pool = grp(lambda s: "maxpool" in s); g_head = grp(lambda s: "head_" in s); g_loss = grp(lambda s: "loss_" in s)
g_pack = grp(lambda s: "pack" in s); g_adam = grp(lambda s: "adam" in s)
dom = [float(r["AverageNs"]) / 1e3 for r in ks if "conv3d_fwd_k3_persist<1, true>" in r["Name"]][0]
tr = json.load(open(R + "r03_hbm_traffic.json"))
pk = tr["kernels"]["conv3d_fwd_k3_persist<1, true>"]
kb, nb, totb, grpb = stats("r03_kernel_stats_bf16.csv")
b_conv = grpb(lambda s: ("lp_conv_" in s or "lp_wgrad" in s or "first_" in s) and "upconv" not in s and "pack" not in s)
b_up = grpb(lambda s: ("upconv" in s or "upwg" in s) and "pack" not in s); b_ct = grpb(lambda s: "convt" in s and "pack" not in s)
b_bn = grpb(lambda s: "bn_relu_bwd" in s or "bn_finalize" in s or "bn_bwd_finalize" in s or "channel_sum" in s); b_pool = grpb(lambda s: "maxpool" in s)
b_head = grpb(lambda s: "head_" in s); b_loss = grpb(lambda s: "loss_" in s)
b_rest = totb - b_conv - b_up - b_ct - b_bn - b_pool - b_head - b_loss


def row(name, j, note):
    r = j["roofline"]
    t = r.get("traffic")
    tt = f", counter traffic {t / 1e6:.0f} MB per launch" if t else ""
    return f"| {name} | {j['ms_per_step']:.2f} | {j['value'] / 1e6:.1f} M | `{r['kernel']}` {r['bound']} {r['frac']:.3f}{tt} | {note} |"


def cbn(j):
    c = j.get("cpu_baseline")
    return "" if not c else f"; CPU oracle on the same workload {c['value'] / 1e6:.2f} M voxels/s ({c['cores']} cores), Dice vs CPU ref {c['dice_vs_cpu_ref']:.6f}, max rel output error {c['max_rel_output_err']:.1e}"


rows = "\n".join([
    row("`python bench.py` (headline: `UNet()` 128³ fp32 train step, HIP graph)", d, "`r03_bench_default.json` (the JSON line as printed); round 2: 3.39 ms"),
    row("same under `torch.distributed.run --nproc-per-node 1` (segmented graph chain + RCCL all-reduces on a side stream)", t1,
        f"`r03_bench_torchrun1.json`: {t1['config']['grad_buckets']} gradient buckets of ≤ {t1['config'].get('grad_bucket_bytes')} B, `comm_ms_exposed` {t1['config']['comm_ms_exposed']} ms (one rank: launch cost only; unmeasured on N > 1 hardware), backend {t1['config'].get('comm_backend')}"),
    row("`--mode infer --batch 2` (BASELINE cfg 2: eval-mode forward, BatchNorm from running statistics)", inf, "`r03_bench_infer_b2.json`, per-stage: `r03_stage_table_infer_b2.md`" + cbn(inf)),
    row("`--dtype bf16` (same step, 16-bit activations)", bf, "`r03_bench_bf16.json`; per-stage `r03_stage_table_bf16.md`; round 2: 2.79 ms" + cbn(bf)),
    row("`--dtype f16`", f16, "`r03_bench_f16.json`; round 2: 2.87 ms" + cbn(f16)),
    row("`--model UNetSP --size 192 --dtype bf16` (cfg 4 patch size)", sp192, f"`r03_bench_UNetSP_192_bf16.json`; round 2: 7.25 ms; fp32 at this size: {sp192f['ms_per_step']:.2f} ms (`r03_bench_UNetSP_192_f32.json`)" + cbn(sp192)),
    row("`--model recAE_v2_fixed --size 192 --dtype bf16` (cfg 4 model)", rec192, "`r03_bench_recAE_192_bf16.json`; per-stage `r03_stage_table_recAE_192_bf16.md`"),
    row("`--model UNetSP --size 256 --dtype f16` (cfg 5)", sp256, f"`r03_bench_UNetSP_256_f16.json`; round 2: 15.69 ms; fp32 at this size: {sp256f['ms_per_step']:.2f} ms (`r03_bench_UNetSP_256_f32.json`)" + cbn(sp256)),
    row("`--model recAE_v2_fixed` (k = 5 legacy net, 128³ fp32)", rec, "`r03_bench_recAE_128_f32.json`; per-stage `r03_stage_table_recAE_f32.md` (k = 5 kernels not touched this round)"),
    row("`--model UNet4_2IC` (k = 5, 2 input channels)", ic, "`r03_bench_UNet4_2IC_128_f32.json`"),
])
txt = f"""# profiles/ — measurements (1× MI355X, gfx950, ROCm 7.2)

## Round 3

Headline workload unchanged: `bench.py` default — `UNet()` (1 in, 2 out, i_size 8, 4 blocks), one 128³ fp32 patch per GPU,
train step = `requires_grad_` input → forward (train-mode BN) → Dice + CE → backward → Adam(amsgrad) → grads None
(`ctunet/pytorch/Model.py:343-374`), replayed from a HIP graph.  All files of this round are named `r03_*`; they were
produced by `scripts/final_measure_r03.sh a|b|c` (three `gpurun` calls; this section: `scripts/make_profiles_readme_r03.py`).
Secondary legs carry the model and batch in their `metric` string; `cpu_baseline` of the secondary legs is one timed oracle step
(`--cpu-check-only`) plus the Dice / output-error check of that precision against the fp32 oracle.

| run | ms/step | voxels/s | roofline kernel, bound, fraction | file / note |
|---|---|---|---|---|
{rows}
| CPU oracle (ATen-CPU fp32, {cb['cores']} granted host cores, no checkpoint recompute) | {2097152 / cb['value'] * 1e3:.0f} | {cb['value'] / 1e6:.2f} M | — | `cpu_baseline` of the headline line, kind "port"; reference default `use_checkpoint=True` (+ one recompute forward, emulated): {cb['checkpoint_default']['value'] / 1e6:.2f} M voxels/s; 8 threads: {cb['threads_8']['value'] / 1e6:.2f} M; Dice of the HIP path's hard segmentation vs the oracle's on identical weights / input {cb['dice_vs_cpu_ref']:.7f}, max relative output error {cb['max_rel_output_err']:.1e} |

What the fp32 headline is made of (`r03_kernel_stats.csv` = `rocprofv3 --kernel-trace --stats` of `python bench.py --steps 20
--warmup 5`; per step = total ÷ {n} executions, which include the eagerly launched roofline steps): GPU-busy {tot:.2f} ms/step =
plain convolutions (forward, data gradient, weight gradient with the folded BatchNorm backward, slab reductions) {g_conv:.2f}, fused
up-convolution family {g_up:.2f}, deep-level ConvTranspose {g_ct:.2f}, BatchNorm {g_app + g_red + g_fin:.2f} (backward apply {g_app:.2f} -- round 2:
0.25 --, reduce {g_red:.2f}, the 34 finalize launches {g_fin:.2f}), pooling {g_pool:.2f}, head {g_head:.2f}, loss {g_loss:.2f}, weight packing {g_pack:.2f},
optimizer {g_adam:.2f}.  Average duration of the roofline kernel `conv3d_fwd_k3_persist<1, true>` in that trace: {dom:.1f} µs (HIP events in
`bench.py`: {d['roofline']['avg_launch_ms'] * 1e3:.1f} µs).
The bf16 step (`r03_kernel_stats_bf16.csv`, same command with `--dtype bf16`, ÷ {nb}): GPU-busy {totb:.2f} ms/step = 16-bit
convolutions {b_conv:.2f}, fused up-convolution family {b_up:.2f}, ConvTranspose {b_ct:.2f}, BatchNorm glue {b_bn:.2f}, pooling {b_pool:.2f}, head
{b_head:.2f}, loss {b_loss:.2f}, rest (packing, optimizer, copies, fills) {b_rest:.2f}.
`r03_hbm_traffic.json` — `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` (separate passes, `bench.py --eager`, fp32 →
section `kernels`, `--dtype bf16` → `kernels_bf16`), FETCH_SIZE doubled as the gfx950 guide prescribes: the fp32 roofline kernel
moves {pk['hbm_bytes_per_launch'] / 1e6:.0f} MB per launch ({pk['read_bytes_per_launch'] / 1e6:.0f} read + {pk['write_bytes_per_launch'] / 1e6:.0f} written) against 112 MB
algorithmic; `bench.py`'s `roofline.traffic` is looked up there for the 16-bit legs too.
`r03_stage_table_f32.md`, `_bf16.md`, `_infer_b2.md`, `_recAE_f32.md`, `_recAE_192_bf16.md` — per conv STAGE (width, padded C_in,
padded C_out) → kernel symbol, launches per step, µs per launch, algorithmic TFLOP/s, GB/s and MB per launch, **counter MB per
launch (read + written) and matrix-pipe busy % of that stage**: three extra `rocprofv3 --pmc` passes (FETCH_SIZE, WRITE_SIZE,
SQ_VALU_MFMA_BUSY_CYCLES; kernel trace only) over `scripts/stage_table.py --profiled`, attributed launch by launch (the i-th
timed launch of a kernel family = the i-th dispatch of that family in the pass).
`r03_pmc_roofline_kernel.txt` — SQ counters of the fp32 roofline kernel (`scripts/pmc_roofline_kernel.sh`); `r03_pmc_lp.txt` —
of the 16-bit 8 → 8 forward (`lp_conv_fwd_pair_kernel`), 8 → 8 weight gradient (`lp_wgrad8_kernel`) and 32 → 8 forward at 128³
(`scripts/pmc_lp.sh`).
"""
extra = open(R + "r03_notes.md").read() if os.path.exists(R + "r03_notes.md") else ""
old = open(R + "README.md").read()
old = old[old.index("## Round 2"):]
open(R + "README.md", "w").write(txt + extra + "\n" + old)
print("ok")
