#!/bin/bash
# SQ counters of the extension stage's kernels (headline step only)
set -u
TAG=${1:-ext}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/prof
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 500 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_WAIT_INST_ANY --output-format csv -d $OUT/${TAG}_sq -- python3 $REPO/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-configs --no-sweep --no-nvbowtie-mode --no-cpp-host --no-robust --no-plain-ab --no-traceback > $OUT/${TAG}_sq.json 2> $OUT/${TAG}_sq.err
echo "rc $?"
python3 - $OUT $TAG <<'PY'
import csv, glob, os, sys, collections
out, tag = sys.argv[1], sys.argv[2]
fs = glob.glob(os.path.join(out, tag + "_sq", "**", "*counter_collection.csv"), recursive=True)
per = collections.defaultdict(float); name_of = {}
for row in csv.DictReader(open(fs[0])):
    key = (row["Dispatch_Id"], row["Counter_Name"])
    per[key] += float(row["Counter_Value"]); name_of[row["Dispatch_Id"]] = row["Kernel_Name"]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for (disp, ctr), v in per.items():
    agg[name_of[disp]][ctr].append(v)
with open(os.path.join(out, tag + "_sq.txt"), "w") as f:
    for k, cs in agg.items():
        if not any(x in k for x in ("ungapped", "gap_chance", "band31", "fm_seed_both")):
            continue
        line = k[:80] + " " + " ".join("%s=%.3g" % (c, sum(v) / len(v)) for c, v in sorted(cs.items()))
        print(line); f.write(line + "\n")
PY
