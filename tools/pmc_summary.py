#!/usr/bin/env python3
"""Summarise a rocprofv3 --pmc CSV directory: per kernel name, launches and the mean of each counter."""
import csv, glob, os, sys, collections
d = sys.argv[1]
rows = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = r.get("Kernel_Name", "?")
        k = k.split("(")[0][:70]
        rows[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
print("%-72s %-22s %8s %16s %16s" % ("kernel", "counter", "launches", "mean", "sum"))
for k in sorted(rows):
    for c in sorted(rows[k]):
        v = rows[k][c]
        print("%-72s %-22s %8d %16.1f %16.1f" % (k, c, len(v), sum(v) / len(v), sum(v)))
