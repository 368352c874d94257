#!/usr/bin/env python3
"""A/B of the full-matrix traceback over opposite-mate-shaped jobs (150 bp reads, 500-symbol windows, 15 % with an indel): the
row-restricted DP (default) against NVBIO_ALN_NO_NARROW_TRACEBACK.  python scripts/exp_full_tb.py [jobs]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import __graft_entry__ as ge
amd = ge.load_package()
import bench
J = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1_000_000
M, W, n = 150, 500, 50_000_000
dev = "cuda:0"
genome = bench.make_reference(n, dev, seed=5)
g = torch.Generator(device=dev); g.manual_seed(3)
wb = torch.randint(0, n - W - 8, (J,), device=dev, generator=g, dtype=torch.int64)
off = torch.randint(0, W - M - 8, (J,), device=dev, generator=g, dtype=torch.int64)
j = torch.arange(M, device=dev, dtype=torch.int64)[None, :]
has = torch.rand(J, device=dev, generator=g) < 0.5                       # half of the jobs carry an indel: those reach the DP
ip = torch.randint(5, M - 5, (J,), device=dev, generator=g)[:, None]
ig = torch.randint(1, 4, (J,), device=dev, generator=g)[:, None]
src = torch.where(has[:, None] & (j >= ip), j + ig, j)
sym = bench.genome_symbols(genome, (wb + off)[:, None] + src)
sub = torch.rand(J, M, device=dev, generator=g) < 0.01
sym = torch.where(sub, (sym + 1) % 4, sym)
reads4 = bench.pack4(sym.reshape(-1))
roff = torch.arange(J + 1, device=dev, dtype=torch.int32) * M
sv = amd.GotohScheme(0, 6, 6, -8, -3, -8, -3)
al = amd.make_gotoh_aligner(amd.SEMI_GLOBAL, sv)
for flags in (0, amd.ALN_NO_NARROW_TRACEBACK):
    batch = amd.AlignmentBatch(reads4, 4, roff, genome, 2, wb.to(torch.int32), (wb + W).to(torch.int32), max_read_len=M, algo_flags=flags)
    sc, sk = amd.BatchedAlignmentScore(al).enact(batch, M, W)
    op = amd.BatchedAlignmentTraceback(al)
    ms = []
    for it in range(3):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        out = op.enact(batch, M, W, cigar_stride=16, scores=sc, sinks=sk)
        b.record(); torch.cuda.synchronize()
        ms.append(a.elapsed_time(b))
    ln = out[4].to(torch.int64) & 0xFFFFFFFF
    print("flags %3d: %.2f ms per %d jobs (%.1f %% gapped), checksum %d" % (flags, sorted(ms)[1], J, float((ln > 1).float().mean()) * 100,
          int((out[3].to(torch.int64) & 0xFFFF).sum())), flush=True)
