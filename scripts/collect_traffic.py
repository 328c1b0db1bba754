"""Turn two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) of bench.py into per-launch HBM traffic per kernel.

    rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_fetch -- python bench.py --eager ...
    rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_write -- python bench.py --eager ...
    python scripts/collect_traffic.py gpurun_out/pmc_fetch gpurun_out/pmc_write profiles/r01_hbm_traffic.json

Corrections as MI355X_MICROARCH.md "HBM" prescribes: FETCH_SIZE/WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE
reports exactly half of the bytes of wide coalesced streaming reads, so it is doubled; WRITE_SIZE is exact for
16-byte-per-lane streaming stores."""
import collections, csv, glob, json, sys


def per_kernel(d, counter):
    f = (glob.glob(f"{d}/*counter_collection.csv") + glob.glob(f"{d}/*/*counter_collection.csv"))[0]
    agg = collections.defaultdict(lambda: [0.0, 0])
    disp = collections.defaultdict(float)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != counter:
            continue
        disp[(r["Dispatch_Id"], r["Kernel_Name"])] += float(r["Counter_Value"])     # sum over XCD instances
    for (_, name), v in disp.items():
        name = name.replace("(anonymous namespace)::", "").replace("void ", "")
        name = name.split("(")[0]
        agg[name][0] += v
        agg[name][1] += 1
    return {k: v[0] / v[1] for k, v in agg.items()}, {k: v[1] for k, v in agg.items()}


fetch, n = per_kernel(sys.argv[1], "FETCH_SIZE")
write, _ = per_kernel(sys.argv[2], "WRITE_SIZE")
out = {}
for k in sorted(fetch):
    rd = fetch[k] * 1024 * 2            # KiB -> B, gfx950 x2 correction
    wr = write.get(k, 0.0) * 1024
    out[k] = {"launches": n[k], "read_bytes_per_launch": rd, "write_bytes_per_launch": wr, "hbm_bytes_per_launch": rd + wr}
json.dump({"note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) on `python bench.py --eager`; "
                   "FETCH_SIZE x2 (gfx950), KiB -> bytes; average per launch over all launches of the symbol",
           "kernels": out}, open(sys.argv[3], "w"), indent=1)
for k, v in sorted(out.items(), key=lambda kv: -kv[1]["hbm_bytes_per_launch"] * kv[1]["launches"])[:12]:
    print(f"{k:50s} n={v['launches']:5d} rd={v['read_bytes_per_launch']/1e6:9.2f} MB wr={v['write_bytes_per_launch']/1e6:9.2f} MB")
