#!/usr/bin/env python
"""Average duration (us) of the LAST n launches of every kernel whose name contains one of the filters, from a rocprofv3
kernel-trace csv:  python tools/probe/trace_avg.py <s_kernel_trace.csv> <n> [filter ...]"""
import csv, sys, collections
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
n = int(sys.argv[2]); flt = sys.argv[3:]
d = collections.defaultdict(list)
for r in rows:
    k = r["Kernel_Name"].split("(")[0].replace("void ", "")
    if not flt or any(f in k for f in flt):
        d[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1][-n:])):
    print(f"{k[:60]:60s} calls {len(v):4d}  avg of last {min(n, len(v)):3d}: {sum(v[-n:]) / len(v[-n:]):8.1f} us")
