#!/usr/bin/env python3
"""Print the kernel sequence of the LAST timed step of a rocprofv3 --kernel-trace CSV of bench.py: every launch between the last two
first-strand seed-pass launches (name, start offset in ms, duration in ms), and the gaps between launches.  Usage:
    trace_step.py <kernel_trace.csv> [marker-substring]"""
import csv
import sys

path = sys.argv[1]
marker = sys.argv[2] if len(sys.argv) > 2 else "fm_seed_pipe_kernel"
rows = []
with open(path) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
marks = [i for i, r in enumerate(rows) if marker in r[2]]
# a step launches the marker twice (fw, rc): the last step starts at the second-to-last marker
if len(marks) < 4:
    sys.exit("marker not found often enough")
lo, hi = marks[-4], marks[-2]
t0 = rows[lo][0]
busy = 0
prev_end = t0
for s, e, n in rows[lo:hi]:
    n = n.replace("nvbio_amd::", "").replace("(anonymous namespace)::", "")
    print("%9.3f  %8.3f  gap %7.3f  %s" % ((s - t0) / 1e6, (e - s) / 1e6, (s - prev_end) / 1e6, n[:150]))
    busy += e - s
    prev_end = max(prev_end, e)
print("step wall %.3f ms, kernel busy %.3f ms, launches %d" % ((rows[hi][0] - t0) / 1e6, busy / 1e6, hi - lo))
