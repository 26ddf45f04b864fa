"""Average every counter of a rocprofv3 --pmc counter_collection.csv over the launches of kernels matching a substring."""
import csv, sys, collections
path, sub = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(list)
for r in csv.DictReader(open(path)):
    if sub in r["Kernel_Name"]:
        acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in sorted(acc.items()):
    print(f"{k:40s} n={len(v):3d} mean={sum(v)/len(v):.4g}")
