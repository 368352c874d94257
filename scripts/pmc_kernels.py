#!/usr/bin/env python3
"""Mean of every counter per kernel from a rocprofv3 --pmc counter_collection.csv (kernel names cut to 100 characters).
Usage: pmc_kernels.py <counter_collection.csv> [substring ...]"""
import collections
import csv
import sys

agg = collections.defaultdict(lambda: collections.defaultdict(list))
for row in csv.DictReader(open(sys.argv[1])):
    agg[row["Kernel_Name"][:100]][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, d in sorted(agg.items()):
    if len(sys.argv) > 2 and not any(s in k for s in sys.argv[2:]):
        continue
    print(k)
    for c, v in sorted(d.items()):
        print("    %-28s mean %16.1f  launches %d" % (c, sum(v) / len(v), len(v)))
