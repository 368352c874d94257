#!/bin/bash
# occupancy experiment on the one-kernel seed pass: step time against the number of 256-thread workgroups in the grid
# (NVBIO_AMD_SEED_GRID_BLOCKS; 2048 = every CU's 32 wave slots filled once).  Result of round 1 (match ms per launch):
#   512: 9.85   1024: 6.87   2048: 6.6   4096: 5.73   8192: 5.58   16384: 5.44   32768: 5.38   65536: 5.32   131072: 5.35   400000: 5.6
# i.e. saturated from 16 waves per CU on: the pass is bound by what the memory system sustains, not by latency per wave.
for b in ${BLOCKS:-512 1024 2048 4096 8192 16384 32768 65536}; do
  NVBIO_AMD_SEED_GRID_BLOCKS=$b timeout -k 10 200 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-plain-ab --no-traceback 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print($b, d['ms_per_step'], d['stage_ms']['match_fw'], d['stage_ms']['match_rc'])" >> gpurun_out/occ.log || exit 1
done
