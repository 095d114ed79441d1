#!/usr/bin/env python
"""Per-kernel averages of a rocprofv3 --pmc counter_collection.csv:  python tools/probe/pmc_kernels.py <csv> [name filter ...]"""
import csv, collections, sys
agg = collections.defaultdict(lambda: collections.defaultdict(float))
calls = collections.defaultdict(set)
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"].split("(")[0][-48:]
    if len(sys.argv) > 2 and not any(f in k for f in sys.argv[2:]):
        continue
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    calls[k].add(r["Dispatch_Id"])
for k, v in sorted(agg.items()):
    n = len(calls[k])
    print(k, f"calls={n}", {c: round(x / n) for c, x in sorted(v.items())})
