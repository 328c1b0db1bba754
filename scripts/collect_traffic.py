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


def section(fetch_dir, write_dir):
    fetch, n = per_kernel(fetch_dir, "FETCH_SIZE")
    write, _ = per_kernel(write_dir, "WRITE_SIZE")
    out = {}
    for k in sorted(fetch):
        rd = fetch[k] * 1024 * 2            # KiB -> B, gfx950 x2 correction
        wr = write.get(k, 0.0) * 1024
        out[k] = {"launches": n[k], "read_bytes_per_launch": rd, "write_bytes_per_launch": wr, "hbm_bytes_per_launch": rd + wr}
    return out


# old form: fetch_dir write_dir out.json; new form: out.json label=fetch_dir,write_dir ... (label "f32" -> "kernels",
# any other label L -> "kernels_L": the 16-bit legs run different symbols, some of which rocprofv3 leaves mangled)
if "=" in "".join(sys.argv[2:]):
    dst, pairs = sys.argv[1], [a.split("=", 1) for a in sys.argv[2:]]
else:
    dst, pairs = sys.argv[3], [("f32", sys.argv[1] + "," + sys.argv[2])]
doc = {"note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) on `python bench.py --eager [--dtype L]`; "
               "FETCH_SIZE x2 (gfx950), KiB -> bytes; average per launch over all launches of the symbol"}
for label, dirs in pairs:
    out = section(*dirs.split(","))
    doc["kernels" if label == "f32" else "kernels_" + label] = out
    print(label)
    for k, v in sorted(out.items(), key=lambda kv: -kv[1]["hbm_bytes_per_launch"] * kv[1]["launches"])[:12]:
        print(f"{k[:60]:60s} n={v['launches']:5d} rd={v['read_bytes_per_launch']/1e6:9.2f} MB wr={v['write_bytes_per_launch']/1e6:9.2f} MB")
json.dump(doc, open(dst, "w"), indent=1)
