"""Per-kernel mean of every PMC counter in a rocprofv3 --pmc output directory (csv)."""
import csv, glob, sys, collections
d = sys.argv[1]
f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for fn in f:
    for r in csv.DictReader(open(fn)):
        acc[r["Kernel_Name"][:44]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in sorted(acc.items()):
    if not k.startswith(("void s5", "s5::")):
        continue
    print(k, " ".join(f"{c}={sum(v)/len(v):.4g}" for c, v in sorted(cs.items())), f"n={len(next(iter(cs.values())))}")
