#!/bin/bash
# A/B runs of bench.py on the headline workload with the round's final code (run through gpurun); one JSON line per variant under
# gpurun_out/abf_*.json -> profiles/r02_variants_final.jsonl.  Each run builds the 3 Gbp index once and times 6 steps.
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out
FAST="--steps 6 --warmup 2 --no-cpu-baseline --no-traceback --no-plain-ab"
run() { tag=$1; shift; timeout -k 10 280 python3 $REPO/bench.py $FAST "$@" > $OUT/abf_$tag.json 2> $OUT/abf_$tag.err || { echo "$tag FAILED"; tail -2 $OUT/abf_$tag.err; return 1; }; echo "$tag done: $(python3 -c "import json,sys; d=json.loads([l for l in open('$OUT/abf_$tag.json') if l.startswith('{')][0]); print(round(d['ms_per_step'],3), {k:round(v,2) for k,v in d['stage_ms'].items()})")"; }
run default &&
run three_waves --pk-three-waves &&
run narrow_entries --no-wide-table &&
run per_strand --no-canonical &&
run with_tb --with-traceback &&
run no_second_chance --algo-flags 128 &&
run no_third_chance --algo-flags 2 &&
run dp_everything --algo-flags 1 &&
run k15 --kmer 15 &&
run repeats --repeat-family 10000 --max-seed-hits 16
