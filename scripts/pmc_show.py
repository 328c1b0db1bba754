"""dev: print the per-kernel averages of the counters collected by scripts/pmc_conv.sh <tag>."""
import csv, glob, collections, sys
T = sys.argv[1]; pat = sys.argv[2] if len(sys.argv) > 2 else "conv3d"
for P in "ABC":
    fs = glob.glob(f"gpurun_out/pmc{P}_{T}/**/*counter_collection.csv", recursive=True)
    if not fs:
        continue
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(fs[0])):
        if pat in r["Kernel_Name"] and "reduce" not in r["Kernel_Name"]:
            acc[r["Kernel_Name"][:70]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, d in acc.items():
        print(k)
        for c, v in d.items():
            print(f"    {c:32s} {sum(v) / len(v):16.0f}  (n={len(v)})")
