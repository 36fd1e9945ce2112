"""Per-kernel averages of a rocprofv3 --pmc csv (counter_collection.csv)."""
import csv, sys, collections, glob
path = sys.argv[1]
files = glob.glob(path + "/**/*counter_collection.csv", recursive=True)
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in files:
    for row in csv.DictReader(open(f)):
        k = row.get("Kernel_Name") or row.get("kernel_name")
        acc[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, d in acc.items():
    if len(sys.argv) > 2 and sys.argv[2] not in k:
        continue
    print(k[:90])
    for c, v in sorted(d.items()):
        print(f"    {c:32s} n={len(v):3d} avg={sum(v)/len(v):.4g}")
