"""Summarise tools/pmc_conv.sh output: per variant, counters of the gather-GEMM kernel averaged over its launches."""
import collections, csv, glob, os, sys
root = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/pmc_conv"
for d in sorted(glob.glob(os.path.join(root, "*_p*"))):
    if not os.path.isdir(d):
        continue
    files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in files:
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if "gather_gemm" in k:
                agg[k[:90]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        print(os.path.basename(d), k)
        for c, vals in sorted(v.items()):
            print(f"     {c:28s} n={len(vals):3d} mean={sum(vals) / len(vals):.4g}")
