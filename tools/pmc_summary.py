"""Summarise rocprofv3 --pmc counter_collection.csv per kernel (mean over dispatches)."""
import csv, sys, collections
for path in sys.argv[1:]:
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    with open(path) as f:
        for row in csv.DictReader(f):
            name = row["Kernel_Name"].split("(")[0].replace("void ", "")[:60]
            agg[name][row["Counter_Name"]].append(float(row["Counter_Value"]))
    for k, cs in agg.items():
        if not k.startswith("fa::"):
            continue
        print(k)
        for c, v in cs.items():
            print(f"   {c:28s} n={len(v):3d} mean={sum(v)/len(v):.4g}")
