#!/usr/bin/env python3
"""scripts/bench_pe.py -- BASELINE.json configs[4] shape on one GPU: P pairs of 2 x 150 bp (FR, insert N(350,50)
clipped to [160,500], 1 % substitutions, 0.1 %/base indels) against the 3 Gbp index; each mate anchored in turn
(seed-and-extend), the other scored by full-matrix DP inside nvBowtie's opposite-mate window, best pair per read
(nvbio-gpl_amd/pipeline.py:paired_end), both mates of the chosen pair traced back to CIGARs.  Not a bench.py line; numbers go into DESIGN.md.

    python scripts/bench_pe.py [pairs] [ref_len]
"""
import importlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    import bench
    import __graft_entry__ as ge
    amd = ge.load_package()
    pipeline = importlib.import_module("nvbio_gpl_amd.pipeline")
    amd.DEFAULT_ALGO_FLAGS = int(os.environ.get("PE_ALGO_FLAGS", "0"))           # A/B: NVBIO_ALN_* switches of the DP calls (results never change)
    dev = "cuda:0"
    P = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10_000_000
    n = int(float(sys.argv[2])) if len(sys.argv) > 2 else 3_000_000_000
    M = 150
    genome = bench.make_reference(n, dev, seed=1234)
    # the canonical two-strand table (16-byte entries) unless PE_TABLE_FLAGS says otherwise (0: the per-strand direct table)
    fmi = amd.FMIndex.build(genome, n, kmer_len=int(os.environ.get("PE_KMER", "17")), sa_int=1,
                            table_flags=int(os.environ.get("PE_TABLE_FLAGS", str(amd.FM_TABLE_CANONICAL_WIDE))))
    g = torch.Generator(device=dev); g.manual_seed(77)
    ins = torch.clamp((torch.randn(P, device=dev, generator=g) * 50 + 350).round().to(torch.int64), 160, 500)
    # mate 1 forward at the fragment start, mate 2 reverse-complemented at its end; make_reads draws loci itself, so
    # draw each mate with it at a forced position by regenerating from the same fragment origin
    left = torch.randint(0, n - 520, (P,), device=dev, generator=g, dtype=torch.int64)

    def mate(pos, seed):
        gg = torch.Generator(device=dev); gg.manual_seed(seed)
        j = torch.arange(M, device=dev, dtype=torch.int64)[None, :]
        out = []
        for b in range(0, P, 1_000_000):
            p = pos[b:b + 1_000_000]
            r = p.numel()
            has = torch.rand(r, device=dev, generator=gg) < (1.0 - (1.0 - 0.001) ** M)
            ip = torch.randint(5, M - 5, (r,), device=dev, generator=gg)[:, None]
            ig = torch.randint(1, 4, (r,), device=dev, generator=gg)[:, None]
            is_del = (torch.rand(r, device=dev, generator=gg) < 0.5)[:, None]
            hasc = has[:, None]
            src = torch.where(hasc & is_del & (j >= ip), j + ig, j)
            src = torch.where(hasc & ~is_del & (j >= ip + ig), j - ig, src)
            sym = bench.genome_symbols(genome, p[:, None] + src)
            rnd = torch.randint(0, 4, (r, M), device=dev, generator=gg, dtype=torch.uint8)
            sym = torch.where(hasc & ~is_del & (j >= ip) & (j < ip + ig), rnd, sym)
            sub = torch.rand(r, M, device=dev, generator=gg) < 0.01
            out.append(torch.where(sub, (sym + 1 + rnd % 3) % 4, sym))
        return torch.cat(out)

    m1 = mate(left, 1)
    m2 = 3 - mate(left + ins - M, 2).flip(1)
    swap = torch.rand(P, device=dev, generator=g) < 0.5
    m1s = torch.where(swap[:, None], m2, m1); m2s = torch.where(swap[:, None], m1, m2)
    b1 = pipeline.ReadBatch(bench.pack4(m1s.reshape(-1)), P, M)
    b2 = pipeline.ReadBatch(bench.pack4(m2s.reshape(-1)), P, M)
    del m1, m2, m1s, m2s
    params = pipeline.SeedExtendParams.end_to_end()
    res = []
    for it in range(3):
        timers = {}
        torch.cuda.synchronize(); t0 = time.perf_counter()
        out = pipeline.paired_end(fmi, genome, n, b1, b2, params, timers=timers, cigar_stride=16)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        stage = {}
        for k, v in timers.items():
            stage[k] = sum(a.elapsed_time(b) for a, b in v)
        res.append((dt, stage))
    dt, stage = sorted(res, key=lambda x: x[0])[1]
    paired = out["anchor"] >= 0
    conc = paired & (out["rc1"] != out["rc2"]) & ((out["pos1"] - out["pos2"]).abs() <= 500)
    true1 = torch.where(swap, left + ins, left + M)                    # end position of mate 1's true alignment
    near = (out["pos1"] - true1).abs() <= 40
    opp_ms = stage.get("opposite_a0", 0.0) + stage.get("opposite_a1", 0.0)
    tb_ms = stage.get("traceback_a0", 0.0) + stage.get("traceback_a1", 0.0)
    cells = 2.0 * P * M * 500
    print(json.dumps({"config": "5 paired-end 2 x 150 bp vs 3 Gbp, FR, insert N(350,50), 1 GPU", "pairs": P, "ms": dt * 1e3,
                      "pairs_per_s": P / dt, "paired_fraction": float(paired.float().mean()),
                      "concordant_fraction": float(conc.float().mean()), "mate1_at_true_locus": float((paired & near).float().mean()),
                      "opposite_mate_full_dp_ms": opp_ms, "traceback_both_mates_ms": tb_ms,
                      "mean_cigar_runs": [float((out["cigar_lens1"].to(torch.int64) & 0xFFFFFFFF).float().mean()),
                                          float((out["cigar_lens2"].to(torch.int64) & 0xFFFFFFFF).float().mean())],
                      "opposite_mate_effective_gcups": cells / (opp_ms * 1e-3) / 1e9 if opp_ms else None,
                      "algo_flags": amd.DEFAULT_ALGO_FLAGS, "stage_ms": stage}), flush=True)


if __name__ == "__main__":
    main()
