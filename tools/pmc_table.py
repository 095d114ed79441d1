#!/usr/bin/env python
"""Summarise rocprofv3 --pmc CSV output (one or more *_counter_collection.csv) into a per-kernel table of mean counter values:
    python tools/pmc_table.py 'gpurun_out/pmc_c*/**/*counter_collection.csv' [kernel-name-substring ...]"""
import csv, glob, sys
from collections import defaultdict

def main(pattern, subs):
    acc = defaultdict(lambda: defaultdict(list))
    for f in glob.glob(pattern, recursive=True):
        for row in csv.DictReader(open(f)):
            name = row["Kernel_Name"].split("(")[0]
            if subs and not any(s in name for s in subs):
                continue
            acc[name][row["Counter_Name"]].append(float(row["Counter_Value"]))
    for name, cs in sorted(acc.items()):
        print(f"== {name}")
        for c, v in sorted(cs.items()):
            print(f"   {c:32s} mean {sum(v) / len(v):16.1f}   n={len(v)}")

if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2:])
