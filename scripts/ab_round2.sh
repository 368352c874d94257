#!/bin/bash
# Round-2 A/B and left-out-cost runs of bench.py on the headline workload (run through gpurun); one JSON line per variant
# under gpurun_out/ab_*.json.  Each run builds the 3 Gbp index once (1-10 s) and times 5 steps.
set -e
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out
FAST="--steps 5 --warmup 2 --no-cpu-baseline --no-traceback"
run() { tag=$1; shift; timeout -k 10 280 python3 $REPO/bench.py $FAST "$@" > $OUT/ab_$tag.json 2> $OUT/ab_$tag.err; echo "$tag done: $(python3 -c "import json,sys; d=json.loads([l for l in open('$OUT/ab_$tag.json') if l.startswith('{')][0]); print(round(d['ms_per_step'],3), {k:round(v,2) for k,v in d['stage_ms'].items()})")"; }
run base
run three_waves --pk-three-waves
run with_tb --with-traceback
run breakdown --build-breakdown --no-plain-ab
run k16 --kmer 16 --no-plain-ab
run repeats --repeat-family 10000 --max-seed-hits 16 --no-plain-ab
