#!/usr/bin/env python3
"""Print the kernels of a rocprofv3 kernel_trace.csv around the last-but-one launch of a marker kernel:
trace_window.py <kernel_trace.csv> <marker> [before] [after]"""
import csv
import sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
marker = sys.argv[2]
before = int(sys.argv[3]) if len(sys.argv) > 3 else 12
after = int(sys.argv[4]) if len(sys.argv) > 4 else 40
sel = [i for i, r in enumerate(rows) if marker in r["Kernel_Name"]]
i0 = sel[-2] if len(sel) > 1 else sel[-1]
t0 = int(rows[i0]["Start_Timestamp"])
for r in rows[max(i0 - before, 0):i0 + after]:
    print("%9.2f %8.3f  %s" % ((int(r["Start_Timestamp"]) - t0) / 1e6, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6,
                              r["Kernel_Name"][:110].replace("nvbio_amd::", "").replace("(anonymous namespace)::", "")))
