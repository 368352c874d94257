#!/bin/bash
# per-kernel durations of the HEADLINE step only (no robust batch, no plain-operator A/B, no traceback): the extension stage's launches
set -u
TAG=${1:-ext}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/prof
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_stats -- python3 $REPO/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-configs --no-sweep --no-nvbowtie-mode --no-cpp-host --no-robust --no-plain-ab --no-traceback > $OUT/${TAG}.json 2> $OUT/${TAG}.err
echo "rc $?"
python3 - $OUT $TAG <<'PY'
import csv, glob, os, sys
out, tag = sys.argv[1], sys.argv[2]
fs = glob.glob(os.path.join(out, tag + "_stats", "**", "*kernel_stats.csv"), recursive=True)
rows = list(csv.DictReader(open(fs[0])))
keep = [r for r in rows if any(k in r["Name"] for k in ("ungapped", "gotoh", "fm_seed", "DevicePartition", "partition", "select", "candidate", "windows", "mapq", "unpack", "job_", "scan", "Scan", "gap_chance"))]
with open(os.path.join(out, tag + "_kernels.txt"), "w") as f:
    for r in keep:
        line = "%-150s calls %5s avg_ms %9.4f total_ms %9.3f" % (r["Name"][:150], r["Calls"], float(r["AverageNs"]) / 1e6, float(r["TotalDurationNs"]) / 1e6)
        print(line); f.write(line + "\n")
import shutil
shutil.copy(fs[0], os.path.join(out, tag + "_kernel_stats.csv"))
PY
