#!/bin/bash
# Round 3: what caps the random-gather rate?  Runs scripts/ubench/gather2 (built in the dev container, travels with the snapshot) as a
# sweep, then the configurations named below under rocprofv3 --pmc, one counter group per pass (never combined with a trace domain).
# Output: gpurun_out/gather2/sweep.jsonl, gpurun_out/gather2/pmc.jsonl (one line per configuration and pass: mean of every counter
# over the launches of gather2_kernel).
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/gather2
BIN=$REPO/scripts/ubench/gather2
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
if [ "${1:-}" = "modes-only" ]; then rc=0; else timeout -k 10 300 $BIN sweep $OUT/sweep.jsonl 2> $OUT/sweep.txt; rc=$?; fi
echo "sweep rc $rc"
if [ $rc -ne 0 ]; then tail -5 $OUT/sweep.txt; exit $rc; fi
[ "${1:-}" = "sweep-only" ] && exit 0

# memory kind x load flavour over 128 GiB: rate and the size of the fabric requests (one pass: three TCC counters)
: > $OUT/modes.jsonl
for mem in 0 1 2; do
  for load in 0 1 2 3; do
    d=$OUT/modes_m${mem}_l${load}
    rm -rf $d
    timeout -k 10 180 rocprofv3 --pmc TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum --output-format csv -d $d -- $BIN one 37 8 4 1 8 0 $mem 6 $load > $d.out 2> $d.err
    rc=$?
    if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "modes $mem $load timed out: stopping"; exit $rc; fi
    f=$(find $d -name "*counter_collection.csv" 2>/dev/null | head -1)
    if [ $rc -ne 0 ] || [ -z "$f" ]; then echo "modes $mem $load failed (rc $rc)"; tail -3 $d.err; else
      python3 - "$f" "$(cat $d.out)" >> $OUT/modes.jsonl <<'EOF2'
import csv, sys, json, collections
agg = collections.defaultdict(list)
for row in csv.DictReader(open(sys.argv[1])):
    if "gather2_kernel" in row["Kernel_Name"]:
        agg[row["Counter_Name"]].append(float(row["Counter_Value"]))
run = json.loads(sys.argv[2])
run["requests_per_launch"] = {k: sum(v) / len(v) for k, v in agg.items()}
print(json.dumps(run))
EOF2
    fi
    rm -rf $d
    # the same un-profiled
    timeout -k 10 120 $BIN one 37 8 4 1 8 0 $mem 6 $load >> $OUT/modes_plain.jsonl 2>> $OUT/modes_plain.txt
    rc=$?
    if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "modes (plain) $mem $load timed out: stopping"; exit $rc; fi
  done
done
echo "modes done"
[ "${1:-}" = "modes-only" ] && exit 0

PMCSETS=(
 "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_LEVEL_sum TCC_BUSY_sum TCC_CYCLE_sum"
 "TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_EA0_RDREQ_GMI_CREDIT_STALL_sum TCC_EA0_RDREQ_IO_CREDIT_STALL_sum TCC_TAG_STALL_sum"
 "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_DRAM_sum GRBM_GUI_ACTIVE"
 "TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum"
 "TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_TRANSLATION_MISS_UNDER_MISS_sum"
 "TCP_UTCL1_STALL_INFLIGHT_MAX_sum TCP_UTCL1_STALL_MULTI_MISS_sum TCP_UTCL1_SERIALIZATION_STALL_sum TCP_UTCL1_STALL_UTCL2_REQ_OUT_OF_CREDITS_sum"
 "TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum"
 "TCP_GATE_EN1_sum TCP_TCP_TA_ADDR_STALL_CYCLES_sum TCP_UTCL1_LFIFO_FULL_sum TCP_CLIENT_UTCL1_INFLIGHT_sum"
 "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_VMEM SQ_INST_LEVEL_VMEM"
)
# flog eb inf chain wps pre uncached
CONFIGS=(
 "37 8 1 1 8 0 0"
 "37 8 4 1 8 0 0"
 "37 8 1 3 8 0 0"
 "30 8 1 1 8 0 0"
 "27 8 1 1 8 0 0"
 "27 8 4 1 8 0 0"
)
: > $OUT/pmc.jsonl
ci=0
for cfg in "${CONFIGS[@]}"; do
  gi=0
  for grp in "${PMCSETS[@]}"; do
    d=$OUT/pmc_c${ci}_g${gi}
    rm -rf $d
    timeout -k 10 180 rocprofv3 --pmc $grp --output-format csv -d $d -- $BIN one $cfg 6 > $d.out 2> $d.err
    rc=$?
    if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "config $ci group $gi timed out: stopping"; exit $rc; fi
    f=$(find $d -name "*counter_collection.csv" 2>/dev/null | head -1)
    if [ $rc -ne 0 ] || [ -z "$f" ]; then echo "config $ci group $gi failed (rc $rc)"; tail -3 $d.err; else
      python3 - "$f" "$cfg" "$(cat $d.out)" >> $OUT/pmc.jsonl <<'EOF'
import csv, sys, json, collections
agg = collections.defaultdict(list)
for row in csv.DictReader(open(sys.argv[1])):
    if "gather2_kernel" in row["Kernel_Name"]:
        agg[row["Counter_Name"]].append(float(row["Counter_Value"]))
try: run = json.loads(sys.argv[3])
except Exception: run = {}
print(json.dumps({"config": sys.argv[2], "G_gathers_per_s_under_profiler": run.get("G_gathers_per_s"), "ms": run.get("ms"),
                  "counters_mean_per_launch": {k: sum(v) / len(v) for k, v in agg.items()}, "launches": {k: len(v) for k, v in agg.items()}}))
EOF
    fi
    rm -rf $d
    gi=$((gi+1))
  done
  echo "config $ci done"
  ci=$((ci+1))
done
echo "all done"
