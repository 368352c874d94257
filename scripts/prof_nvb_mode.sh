#!/bin/bash
# kernel totals of bench.py's nvbowtie_mode leg (the C++ best-approx host loop over 10 M reads): everything between the last best_approx_init_kernel and the end
set -u
TAG=${1:-nvb}
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out/prof
cd /tmp && export TMPDIR=/tmp
timeout -k 10 600 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/prof/${TAG}_trace -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-sweep --no-configs --no-cpp-host --no-robust --no-plain-ab --no-traceback > $R/gpurun_out/prof/${TAG}.json 2> $R/gpurun_out/prof/${TAG}.err
echo "rc $?"
python3 - $(find $R/gpurun_out/prof/${TAG}_trace -name "*kernel_trace.csv" | head -1) $R/gpurun_out/prof/${TAG}_loop_kernels.txt <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
inits = [i for i, r in enumerate(rows) if 'best_approx_init_kernel' in r['Kernel_Name']]
a = inits[-1]
last = max(i for i, r in enumerate(rows) if 'score_reduce_effort' in r['Kernel_Name'])
sel = rows[a:last + 1]
span = (int(sel[-1]['End_Timestamp']) - int(sel[0]['Start_Timestamp'])) / 1e6
busy = sum(int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in sel) / 1e6
agg = collections.defaultdict(lambda: [0, 0.0])
for r in sel:
    k = r['Kernel_Name'].replace('nvbio_amd::', '')[:110]
    agg[k][0] += 1; agg[k][1] += (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6
with open(sys.argv[2], 'w') as f:
    f.write("# span %.2f ms, kernels busy %.2f ms, %d launches\n" % (span, busy, len(sel)))
    for k, v in sorted(agg.items(), key=lambda x: -x[1][1])[:30]:
        f.write("%6d %9.3f  %s\n" % (v[0], v[1], k))
print(open(sys.argv[2]).read())
PY
