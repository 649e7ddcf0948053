"""Summarise a rocprofv3 --pmc counter_collection.csv per kernel (mean over dispatches)."""
import collections
import csv
import sys

d = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][-48:]
    d[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
flt = sys.argv[2] if len(sys.argv) > 2 else ""
for k, v in d.items():
    if flt in k:
        print(k, {c: f"{sum(x) / len(x):.4g}" for c, x in sorted(v.items())}, "n=%d" % len(next(iter(v.values()))))
