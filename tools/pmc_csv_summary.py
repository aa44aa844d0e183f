"""Per-kernel averages of a rocprofv3 --pmc counter_collection.csv: usage pmc_csv_summary.py file.csv [kernel substring]"""
import collections
import csv
import sys
rows = list(csv.DictReader(open(sys.argv[1])))
pat = sys.argv[2] if len(sys.argv) > 2 else ""
agg = collections.defaultdict(lambda: collections.defaultdict(float))
disp = collections.defaultdict(set)
for r in rows:
    k = r["Kernel_Name"][:70]
    if pat not in k:
        continue
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    disp[k].add(r["Dispatch_Id"])
for k, v in agg.items():
    n = len(disp[k])
    print(k, "dispatches", n)
    for c, val in sorted(v.items()):
        print(f"    {c:28s} {val / n:14.1f}")
