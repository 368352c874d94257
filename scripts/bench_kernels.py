#!/usr/bin/env python3
"""scripts/bench_kernels.py -- kernel-level timings at the other BASELINE.json configurations
(SURVEY.md 8d): config 1 (sw-benchmark shape: 100 k x 100 bp patterns vs one 4,096-symbol text,
full-matrix GLOBAL Gotoh(2,-1,-2,-1), text blocking), config 2 (1 M x 22 bp seeds vs the 3 Gbp
index, 90 % substrings + 10 % random) and config 4's per-GPU slice (6.25 M pairs of 150 x 181,
band 31 LOCAL, match 2 / mismatch -6 / gaps -8,-3).  These are not bench.py lines; the numbers go
into DESIGN.md.  Parity of the same shapes is tests/test_gpu_gotoh_full.py and tests/test_gpu_fullsize.py.

    python scripts/bench_kernels.py [--skip-index]     (one JSON line per configuration)
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def timed(torch, fn, reps=5, warm=1):
    for _ in range(warm):
        fn()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, b in ev:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    t = sorted(a.elapsed_time(b) for a, b in ev)
    return t[len(t) // 2]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--skip-index", action="store_true", help="skip the configurations that need the 3 Gbp index")
    args = ap.parse_args()
    import numpy as np
    import torch
    import bench
    import __graft_entry__ as ge
    amd = ge.load_package()
    dev = "cuda:0"

    # ---- config 1: sw-benchmark shape --------------------------------------------------------------
    R, M, N = 100_000, 100, 4096
    g = torch.Generator(device=dev); g.manual_seed(1)
    pats = torch.randint(0, 4, (R * M,), device=dev, generator=g, dtype=torch.uint8)
    text = torch.randint(-2 ** 31, 2 ** 31 - 1, (N // 16 + 8,), device=dev, generator=g, dtype=torch.int64).to(torch.int32)
    reads4 = bench.pack4(pats)
    roffs = (torch.arange(R + 1, device=dev) * M).to(torch.int32)
    wb = torch.zeros(R, dtype=torch.int32, device=dev); we = torch.full((R,), N, dtype=torch.int32, device=dev)
    batch = amd.AlignmentBatch(reads4, 4, roffs, text, 2, wb, we)
    for name, typ in (("GLOBAL", amd.GLOBAL), ("LOCAL", amd.LOCAL), ("SEMI_GLOBAL", amd.SEMI_GLOBAL)):
        for tb in (True, False):
            op = amd.BatchedAlignmentScore(amd.make_gotoh_aligner(typ, amd.SimpleGotohScheme(2, -1, -2, -1)), text_blocking=tb)
            ms = timed(torch, lambda: op.enact(batch, M, N))
            print(json.dumps({"config": "1 sw-benchmark 100k x 100 x 4096", "type": name,
                              "blocking": "text" if tb else "pattern", "ms": ms,
                              "gcups": R * M * N / (ms * 1e-3) / 1e9}), flush=True)
    # ---- config 5 shape: opposite-mate full DP, 150 bp mates in 400-symbol windows of a synthetic genome ----
    P5, M5, W5, n5 = 2_000_000, 150, 400, 400_000_000
    genome5 = bench.make_reference(n5, dev, seed=5)
    reads5, pos5, rc5 = bench.make_reads(genome5, n5, P5, M5, dev, seed=6)
    reads5 = torch.where(rc5[:, None], 3 - reads5.flip(1), reads5)
    r45 = bench.pack4(reads5.reshape(-1))
    roffs5 = (torch.arange(P5 + 1, device=dev) * M5).to(torch.int32)
    g.manual_seed(7)
    off5 = torch.randint(0, W5 - M5 - 8, (P5,), device=dev, generator=g)
    wb5 = torch.clamp(pos5 - off5, min=0); we5 = torch.clamp(wb5 + W5, max=n5)
    b5 = amd.AlignmentBatch(r45, 4, roffs5, genome5, 2, wb5.to(torch.int32), we5.to(torch.int32), max_read_len=M5)
    for name, typ, sv in (("SEMI_GLOBAL e2e (0,-6,-8,-3)", amd.SEMI_GLOBAL, (0, 6, 6, -8, -3, -8, -3)),
                          ("LOCAL (2,-6,-8,-3)", amd.LOCAL, (2, 6, 6, -8, -3, -8, -3))):
        for tb in (True, False):
            op = amd.BatchedAlignmentScore(amd.make_gotoh_aligner(typ, amd.GotohScheme(*sv)), text_blocking=tb)
            ms = timed(torch, lambda: op.enact(b5, M5, W5), reps=3)
            print(json.dumps({"config": "5 opposite-mate full DP 2M x (150 x 400)", "type": name, "blocking": "text" if tb else "pattern",
                              "ms": ms, "gcups": P5 * M5 * W5 / (ms * 1e-3) / 1e9}), flush=True)
    del genome5, reads5, r45, b5
    if args.skip_index:
        return

    # ---- the 3 Gbp reference and its index ---------------------------------------------------------
    n = 3_000_000_000
    genome = bench.make_reference(n, dev, seed=1234)
    fmi = amd.FMIndex.build(genome, n, kmer_len=16, sa_int=1)

    # ---- config 2: 1 M x 22 bp seeds ---------------------------------------------------------------
    Q, L = 1_000_000, 22
    g.manual_seed(2)
    starts = torch.randint(0, n - L, (Q,), device=dev, generator=g, dtype=torch.int64)
    sym = bench.genome_symbols(genome, starts[:, None] + torch.arange(L, device=dev)[None, :])
    rnd = torch.randint(0, 4, (Q, L), device=dev, generator=g, dtype=torch.uint8)
    sym[::10] = rnd[::10]                                              # 10 % random 22-mers (mostly miss)
    padded = torch.zeros((Q, 24), dtype=torch.uint8, device=dev); padded[:, :L] = sym
    q4 = bench.pack4(padded.reshape(-1))
    qs = amd.PackedStringSet(q4, 4, Q, fixed_len=L, stride=24, device=dev)
    for label, flags in (("k=16 table", 0), ("no table (reference algorithm)", amd.FM_NO_KMER_TABLE)):
        ms = timed(torch, lambda: fmi.match(qs, flags), reps=20, warm=3)
        _, blk = fmi.match(qs, amd.FM_NO_KMER_TABLE, want_blocks=True)
        alg = int((blk.to(torch.int64) & 0xFFFFFFFF).sum()) * 32 + Q * (11 + 8)
        print(json.dumps({"config": "2 FM seed pass 1M x 22 vs 3 Gbp", "variant": label, "ms": ms,
                          "queries_per_s": Q / (ms * 1e-3), "algorithmic_GBps": alg / (ms * 1e-3) / 1e9}), flush=True)
    flt = amd.FMIndexFilter()
    total = flt.rank(fmi, qs)
    ms = timed(torch, lambda: flt.locate(0, total), reps=20, warm=3)
    print(json.dumps({"config": "2 locate of every hit", "hits": int(total), "ms": ms, "hits_per_s": total / (ms * 1e-3)}), flush=True)

    # ---- the reference's "approximate search" benchmark shape (nvbio-test/fmindex_test.cu:890-965): hamming_backtrack -------
    if os.environ.get("BENCH_ONLY", "approx") == "approx" or not os.environ.get("BENCH_ONLY"):
        for L2_, seed_, mm_ in ((32, 0, 1), (50, 25, 2)):
            g.manual_seed(3)
            st = torch.randint(0, n - L2_, (Q,), device=dev, generator=g, dtype=torch.int64)
            sy = bench.genome_symbols(genome, st[:, None] + torch.arange(L2_, device=dev)[None, :])
            sub = torch.rand(sy.shape, device=dev, generator=g) < 0.01
            sy = torch.where(sub, (sy + 1) % 4, sy)
            stride = 56
            pad = torch.zeros((Q, stride), dtype=torch.uint8, device=dev); pad[:, :L2_] = sy
            qa = amd.PackedStringSet(bench.pack4(pad.reshape(-1)), 4, Q, fixed_len=L2_, stride=stride, device=dev)
            ms = timed(torch, lambda: fmi.hamming_backtrack(qa, seed_, mm_), reps=5, warm=1)
            cnt, nr, _ = fmi.hamming_backtrack(qa, seed_, mm_)
            print(json.dumps({"config": "FM approximate search (hamming_backtrack) 1M x %d bp, seed %d, <= %d mismatches, vs 3 Gbp" % (L2_, seed_, mm_),
                              "ms": ms, "queries_per_s": Q / (ms * 1e-3), "queries_with_hits": float((cnt != 0).float().mean()),
                              "mean_ranges": float(nr.float().mean())}), flush=True)

    # ---- config 4 slice: 6.25 M pairs, band 31 -----------------------------------------------------
    P, M = 6_250_000, 150
    g.manual_seed(4)
    reads_sym, pos, rc = bench.make_reads(genome, n, P, M, dev, seed=4)
    reads_sym = torch.where(rc[:, None], 3 - reads_sym.flip(1), reads_sym)     # back on the forward strand
    r4 = bench.pack4(reads_sym.reshape(-1))
    roffs = (torch.arange(P + 1, device=dev) * M).to(torch.int32)
    wbeg = torch.clamp(pos - 15, min=0)
    wend = torch.clamp(wbeg + 31 + M, max=n)
    def i32(t):
        return torch.where(t >= 2 ** 31, t - 2 ** 32, t).to(torch.int32)
    for name, typ, sv in (("LOCAL (2,-6,-8,-3)", amd.LOCAL, (2, 6, 6, -8, -3, -8, -3)),
                          ("SEMI_GLOBAL e2e (0,-6,-8,-3)", amd.SEMI_GLOBAL, (0, 6, 6, -8, -3, -8, -3)),
                          ("GLOBAL (0,-6,-8,-3)", amd.GLOBAL, (0, 6, 6, -8, -3, -8, -3))):
        for hint, kern in ((M, "packed int16"), (0, "int32")):
            b = amd.AlignmentBatch(r4, 4, roffs, genome, 2, i32(wbeg), i32(wend), max_read_len=hint)
            al = amd.make_gotoh_aligner(typ, amd.GotohScheme(*sv))
            ms = timed(torch, lambda: amd.batch_banded_alignment_score(31, al, b))
            print(json.dumps({"config": "4 banded extend 6.25M x (150,181) band 31", "type": name, "kernel": kern, "ms": ms,
                              "gcups": P * 31 * M / (ms * 1e-3) / 1e9, "pairs_per_s": P / (ms * 1e-3)}), flush=True)


if __name__ == "__main__":
    main()
