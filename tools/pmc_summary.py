#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc csv output: per kernel name, mean of every counter over dispatches."""
import csv, glob, sys, collections
root = sys.argv[1]
pat = sys.argv[2] if len(sys.argv) > 2 else ""
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"]
        if pat and pat not in k: continue
        agg[(k[:70], row.get("Grid_Size"), row.get("LDS_Block_Size"))][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, cs in agg.items():
    print(k)
    for c, v in sorted(cs.items()):
        print(f"   {c:34s} n={len(v):3d} mean={sum(v)/len(v):16.1f}")
