#!/usr/bin/env python3
"""scripts/exp_traceback.py -- time the banded traceback on P (150, 181) pairs drawn from a synthetic genome
(1 % substitutions, 0.1 %/base indels): shortcut on/off.  Used under rocprofv3 --kernel-trace --stats."""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
import __graft_entry__ as ge
amd = ge.load_package()
dev = "cuda:0"
P, M, n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 2_000_000, 150, 200_000_000
genome = bench.make_reference(n, dev, seed=3)
reads_sym, pos, rc = bench.make_reads(genome, n, P, M, dev, seed=4)
reads_sym = torch.where(rc[:, None], 3 - reads_sym.flip(1), reads_sym)
r4 = bench.pack4(reads_sym.reshape(-1))
roffs = (torch.arange(P + 1, device=dev) * M).to(torch.int32)
wb = torch.clamp(pos - 15, min=0); we = torch.clamp(wb + 31 + M, max=n)
b = amd.AlignmentBatch(r4, 4, roffs, genome, 2, wb.to(torch.int32), we.to(torch.int32), max_read_len=M)
al = amd.make_gotoh_aligner(amd.SEMI_GLOBAL, amd.GotohScheme(0, 6, 6, -8, -3, -8, -3))
op = amd.BatchedBandedAlignmentTraceback(31, al)
for mode in ("shortcut", "dp-only"):
    if mode == "dp-only":
        os.environ["NVBIO_AMD_NO_UNGAPPED_TRACEBACK"] = "1"
    ts = []
    for _ in range(4):
        a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); out = op.enact(b, cigar_stride=16); e.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(e))
    print(json.dumps({"mode": mode, "pairs": P, "ms": ts, "mean_cigar": float((out[4].to(torch.int64) & 0xFFFFFFFF).float().mean())}), flush=True)
