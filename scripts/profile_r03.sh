#!/bin/bash
# Round 3 rocprofv3 evidence for bench.py (run through gpurun):
#   1. kernel trace + stats of the default run (headline step + robust-input step)      -> gpurun_out/prof/<tag>_stats
#   2. PMC passes over the headline step only, each on its own (never combined with a trace domain):
#        fabric read requests by size (the 128-byte line finding), FETCH_SIZE, WRITE_SIZE, SQ issue / wait counters
set -u
TAG=${1:-r03a}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/prof
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
run() {   # name, args..., then -- counters
  name=$1; shift
  timeout -k 10 400 rocprofv3 "$@" > $OUT/${TAG}_${name}.json 2> $OUT/${TAG}_${name}.err
  rc=$?
  echo "$name rc $rc"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "timed out: stopping"; exit $rc; fi
}
FULL="--steps 3 --warmup 1 --no-cpu-baseline --no-configs --no-sweep --no-nvbowtie-mode --no-cpp-host"
FAST="--steps 3 --warmup 1 --no-cpu-baseline --no-configs --no-sweep --no-nvbowtie-mode --no-robust --no-plain-ab --no-traceback"
run stats --kernel-trace --stats --output-format csv -d $OUT/${TAG}_stats -- python3 $REPO/bench.py $FULL
run rdreq --pmc TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum --output-format csv -d $OUT/${TAG}_rdreq -- python3 $REPO/bench.py $FAST
run fetch --pmc FETCH_SIZE --output-format csv -d $OUT/${TAG}_fetch -- python3 $REPO/bench.py $FAST
run write --pmc WRITE_SIZE --output-format csv -d $OUT/${TAG}_write -- python3 $REPO/bench.py $FAST
run sq --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU --output-format csv -d $OUT/${TAG}_sq -- python3 $REPO/bench.py $FAST
run wrreq --pmc TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum --output-format csv -d $OUT/${TAG}_wrreq -- python3 $REPO/bench.py $FAST
run tcc --pmc TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE --output-format csv -d $OUT/${TAG}_tcc -- python3 $REPO/bench.py $FAST
# keep the summaries small: per-kernel means of every counter, the stats csv
python3 - $OUT $TAG <<'PY'
import csv, glob, json, os, sys, collections
out, tag = sys.argv[1], sys.argv[2]
res = {}
for name in ("rdreq", "fetch", "write", "wrreq", "sq", "tcc"):
    fs = glob.glob(os.path.join(out, "%s_%s" % (tag, name), "**", "*counter_collection.csv"), recursive=True)
    if not fs:
        continue
    # a dispatch can report one counter in several rows (one per XCD / dimension instance): sum them per dispatch, then average over dispatches
    per = collections.defaultdict(float); name_of = {}
    for row in csv.DictReader(open(fs[0])):
        key = (row["Dispatch_Id"], row["Counter_Name"])
        per[key] += float(row["Counter_Value"]); name_of[row["Dispatch_Id"]] = row["Kernel_Name"][:140]
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for (did, c), v in per.items():
        agg[name_of[did]][c].append(v)
    for k, d in agg.items():
        for c, v in d.items():
            res.setdefault(k, {})[c] = {"mean": sum(v) / len(v), "launches": len(v)}
json.dump(res, open(os.path.join(out, "%s_pmc.json" % tag), "w"), indent=1)
fs = glob.glob(os.path.join(out, "%s_stats" % tag, "**", "*kernel_stats.csv"), recursive=True)
if fs:
    import shutil
    shutil.copy(fs[0], os.path.join(out, "%s_kernel_stats.csv" % tag))
print("summaries written")
PY
# the raw per-dispatch csv files are large: drop them, keep the summaries
rm -rf $OUT/${TAG}_wrreq $OUT/${TAG}_rdreq $OUT/${TAG}_fetch $OUT/${TAG}_write $OUT/${TAG}_sq $OUT/${TAG}_tcc $OUT/${TAG}_stats
ls -la $OUT | tail -20
