#!/bin/bash
# Collect the rocprofv3 evidence for bench.py on the GPU box (run through gpurun):
#   1. kernel trace + stats (per-kernel time)   -> gpurun_out/prof/<tag>_stats
#   2. PMC pass FETCH_SIZE (HBM read traffic)   -> gpurun_out/prof/<tag>_fetch
#   3. PMC pass WRITE_SIZE                      -> gpurun_out/prof/<tag>_write
# PMC passes run on their own (never combined with trace domains), as the pool requires.
set -e
TAG=${1:-r01}
ARGS=${2:-"--steps 2 --warmup 1 --no-cpu-baseline"}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/prof
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_stats -- python3 $REPO/bench.py $ARGS > $OUT/${TAG}_stats.json 2> $OUT/${TAG}_stats.err
echo "stats done"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/${TAG}_fetch -- python3 $REPO/bench.py $ARGS > $OUT/${TAG}_fetch.json 2> $OUT/${TAG}_fetch.err
echo "fetch done"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/${TAG}_write -- python3 $REPO/bench.py $ARGS > $OUT/${TAG}_write.json 2> $OUT/${TAG}_write.err
echo "write done"
# 4. SQ issue/stall counters (VALU utilisation of the extend kernel) and 5. L2 hit/miss, each in its own pass
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU --output-format csv -d $OUT/${TAG}_sq -- python3 $REPO/bench.py $ARGS > $OUT/${TAG}_sq.json 2> $OUT/${TAG}_sq.err
echo "sq done"
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE --output-format csv -d $OUT/${TAG}_tcc -- python3 $REPO/bench.py $ARGS > $OUT/${TAG}_tcc.json 2> $OUT/${TAG}_tcc.err
echo "tcc done"
# 6. address translation (UTCL1) and 7. fabric read requests by size, each in its own pass
FAST="$ARGS --no-plain-ab --no-traceback"
rocprofv3 --pmc TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_TRANSLATION_MISS_sum --output-format csv -d $OUT/${TAG}_utcl -- python3 $REPO/bench.py $FAST > $OUT/${TAG}_utcl.json 2> $OUT/${TAG}_utcl.err || echo "utcl pass failed"
echo "utcl done"
rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_REQ_sum --output-format csv -d $OUT/${TAG}_ea -- python3 $REPO/bench.py $FAST > $OUT/${TAG}_ea.json 2> $OUT/${TAG}_ea.err || echo "ea pass failed"
echo "ea done"
find $OUT -name "*.csv" | head -50
