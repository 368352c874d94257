#!/usr/bin/env python3
"""Print (calls, average ms, name) of the kernels of a rocprofv3 kernel_stats.csv whose name contains any of the given substrings."""
import csv
import sys
for r in csv.DictReader(open(sys.argv[1])):
    if len(sys.argv) < 3 or any(s in r["Name"] for s in sys.argv[2:]):
        print("%6s  %9.4f ms  %s" % (r["Calls"], float(r["AverageNs"]) / 1e6, r["Name"][:110].replace("nvbio_amd::", "")))
