#!/usr/bin/env python3
"""Averages rocprofv3 --pmc counter CSVs per kernel (one row per dispatch and counter)."""
import csv, glob, os, sys, collections
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(out, "pass*", "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        k = row.get("Kernel_Name", "?").split("(")[0][:60]
        acc[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, cs in sorted(acc.items()):
    if not any(t in k for t in ("assemble", "spmv", "cg_", "state10")):
        continue
    print(k)
    for c, v in sorted(cs.items()):
        print(f"   {c:40s} n={len(v):3d}  mean={sum(v)/len(v):.6g}")
