"""Print per-kernel averages of every counter in a rocprofv3 --pmc csv output directory:  python tools/pmc_dump.py <dir> [substr]"""
import collections, csv, glob, sys
acc = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = (row.get("Kernel_Name") or row.get("kernel_name"))[:60]
        if len(sys.argv) > 2 and sys.argv[2] not in k:
            continue
        acc[(k, row["Counter_Name"])].append(float(row["Counter_Value"]))
for (k, c), v in sorted(acc.items()):
    print(f"{k:60s} {c:28s} n={len(v):3d} avg={sum(v) / len(v):.4g}")
