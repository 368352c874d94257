#!/bin/bash
# kernel trace of a bench run without the plain-operator A/B; prints the kernels of the robust-input phase (everything after the second index build)
TAG=${1:-r03c}
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out/prof
cd /tmp && export TMPDIR=/tmp
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof/${TAG}_stats -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-configs --no-sweep --no-plain-ab --no-traceback --no-cpp-host --no-nvbowtie-mode > $R/gpurun_out/prof/${TAG}.json 2> $R/gpurun_out/prof/${TAG}.err
rc=$?
[ $rc -ne 0 ] && { tail -5 $R/gpurun_out/prof/${TAG}.err; exit $rc; }
python3 - $(find $R/gpurun_out/prof/${TAG}_stats -name "*kernel_trace.csv" | head -1) $R/gpurun_out/prof/${TAG}_robust_kernels.txt $R/gpurun_out/prof/${TAG}_headline_kernels.txt <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'fm_ctab_fill_kernel' in r['Kernel_Name']]
def dump(sel, path, div):
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in sel:
        k = r['Kernel_Name'].replace('nvbio_amd::', '')[:150]
        agg[k][0] += 1; agg[k][1] += (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6
    with open(path, 'w') as f:
        f.write("# calls  total ms  ms per step (total / %d steps)  kernel\n" % div)
        for k, v in sorted(agg.items(), key=lambda x: -x[1][1]):
            if 'ctab' in k or v[1] / div < 0.004: continue
            f.write("%5d %10.3f %9.4f  %s\n" % (v[0], v[1], v[1] / div, k))
# robust phase: after the second table build; its steps = 1 warm + 3 timed
dump(rows[idx[1] + 1:], sys.argv[2], 4)
# headline phase: between the first table build and the second one's start; steps = 1 warm + 3 timed (+ accounting launches)
first_seed = next(i for i, r in enumerate(rows) if i > idx[0] and 'fm_seed_both_kernel' in r['Kernel_Name'])
second_build = next(i for i, r in enumerate(rows) if i > first_seed and 'fm_ctab_count_kernel' in r['Kernel_Name'])
dump(rows[first_seed:second_build], sys.argv[3], 4)
PY
rm -rf $R/gpurun_out/prof/${TAG}_stats
head -45 $R/gpurun_out/prof/${TAG}_robust_kernels.txt
