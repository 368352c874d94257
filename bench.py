#!/usr/bin/env python3
"""bench.py -- the hot path on BASELINE.json's headline configuration.

Workload (config.workload = "nvbowtie-se-150bp-3gbp", BASELINE.json configs[2]): per GPU, one
step = one pass of the seed-and-extend path over a batch of 10 M synthetic 150 bp reads against
a 3 Gbp synthetic reference: 2 x 9 exact 22-mer seeds per read through the FM-index (match +
scan), expand + locate of every hit, candidate loci by diagonal, band-31 Gotoh of every
candidate window (nvBowtie's default end-to-end mode: SEMI_GLOBAL, match 0, mismatch -6 at constant
q >= 40, gaps -8/-3; --mode local for its local mode), best alignment per read.  Inputs (index, genome, reads) are resident in HBM
when the timed region starts.  value = reads/s over all ranks (weak scaling: every rank maps its
own 10 M-read shard against its own replica of the index; one RCCL gather of the per-read best
(score, position) to rank 0 closes each step).

    python bench.py --gpus 1 --steps 3 --warmup 1
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

Prints ONE JSON line on rank 0 (see DESIGN.md "Measurement" for every field).
"""
import argparse
import importlib
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0            # MI355X_MICROARCH.md: HBM3E peak 8.0 TB/s (6.29 TB/s measured copy)
GPU_CLOCK_GHZ = 2.4                  # MI355X peak engine clock (MI355X_MICROARCH.md)
DP_ROW_INSTRUCTIONS = 323            # banded_gotoh_band31_pk_kernel<SEMI_GLOBAL,4,2,true,false,true> (binary16 lanes): instructions of one pass through the row loop
                                     # (profiles/r03_pk_row_loop_f16.s; the int16 build: 353, profiles/r02h_pk_row_loop.s)
SECTOR = 64                      # the unit the kernel's accounting instantiation counts gathers in (distinct 64-byte sectors)
LINE = 128                       # bytes the fabric MOVES for one gather: every L2 miss of gfx950 is a 128-byte read request, whatever
                                 # the load's width, the memory's kind or the cache policy bits (TCC_EA0_RDREQ_128B = gathers, _64B = _32B = 0:
                                 # profiles/r03_gather2_modes.jsonl, r03_gather2_pmc_128GiB.jsonl); FETCH_SIZE tallies them at 64 bytes
TB_STRIDE = 32                   # io::Cigar elements kept per read by the traceback stage
SEED_KERNEL_TAG = "fm_seed_pipe_kernel<4>"            # the per-strand pass (--no-canonical)
SEED_BOTH_KERNEL_TAG = "fm_seed_both_kernel<4, false, %s, %s>"   # the two-strand pass over the canonical table (default): % wide entries, deferred heavy searches


def log(msg):
    if int(os.environ.get("RANK", "0")) == 0:
        print("[bench] " + msg, file=sys.stderr, flush=True)


# ---------------------------------------------------------------------------------------------
# synthetic inputs (generated on the GPU, directly in packed form; seeds fixed)
# ---------------------------------------------------------------------------------------------
def make_reference(n, device, seed):
    """3 Gbp i.i.d. uniform reference as 2-bit big-endian packed words (io::SequenceData<DNA> layout)"""
    import torch
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    nw = (n + 15) // 16
    words = torch.randint(-2 ** 31, 2 ** 31 - 1, (nw + 8,), dtype=torch.int64, device=device, generator=g).to(torch.int32)
    return words


def genome_symbols(words, idx):
    """2-bit symbols at int64 indices idx"""
    import torch
    w = words[idx >> 4].to(torch.int64) & 0xFFFFFFFF
    return ((w >> (30 - 2 * (idx & 15))) & 3).to(torch.uint8)


def pack4(sym_flat):
    """uint8 symbols (0..4) -> 4-bit big-endian packed int32 words (io::SequenceData<DNA_N> layout)"""
    import torch
    n = sym_flat.numel()
    pad = (-n) % 8
    if pad:
        sym_flat = torch.cat([sym_flat, torch.zeros(pad, dtype=torch.uint8, device=sym_flat.device)])
    s = sym_flat.view(-1, 8).to(torch.int64)
    sh = torch.tensor([28, 24, 20, 16, 12, 8, 4, 0], device=sym_flat.device, dtype=torch.int64)
    w = (s << sh[None, :]).sum(dim=1)
    w = torch.where(w >= 2 ** 31, w - 2 ** 32, w)
    return torch.cat([w.to(torch.int32), torch.zeros(4, dtype=torch.int32, device=sym_flat.device)])


def plant_family(words, n, copies, device, seed, unit=300):
    """overwrite `copies` random loci of the reference (16-symbol aligned) with one random `unit`-bp element: a repeat family.
    Returns the loci (int64 tensor)."""
    import torch
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    uw = (unit + 15) // 16
    elem = torch.randint(-2 ** 31, 2 ** 31 - 1, (uw,), dtype=torch.int64, device=device, generator=g).to(torch.int32)
    slots = torch.randperm((n // 16 - uw - 8) // (2 * uw), device=device, generator=g)[:copies].to(torch.int64) * (2 * uw)   # disjoint word offsets
    idx = (slots[:, None] + torch.arange(uw, device=device)[None, :]).view(-1)
    words[idx] = elem.repeat(copies)
    return slots * 16


def make_ragged(reads_sym, min_len, device, seed):
    """cut every read of a [R, M] batch to a length drawn uniformly from [min_len, M] and give every base a quality from Illumina's
    four bins plus 40 (q -> nvBowtie mismatch penalty 2 + int(min(q,40)/40 * 4): 2, 3, 4, 5, 6).  Returns (flat symbols uint8, offsets
    int32 [R+1], qualities uint8, lengths int64)."""
    import torch
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    R, M = reads_sym.shape
    lens = torch.randint(min_len, M + 1, (R,), device=device, generator=g, dtype=torch.int64)
    keep = torch.arange(M, device=device)[None, :] < lens[:, None]
    flat = reads_sym[keep]
    offs = torch.zeros(R + 1, dtype=torch.int64, device=device)
    offs[1:] = torch.cumsum(lens, 0)
    bins = torch.tensor([2, 12, 23, 37, 40], dtype=torch.uint8, device=device)
    cum = torch.tensor([0.04, 0.12, 0.25, 0.70], device=device)                     # P(q) = 0.04, 0.08, 0.13, 0.45, 0.30
    quals = torch.empty(flat.numel(), dtype=torch.uint8, device=device)
    for b0 in range(0, flat.numel(), 1 << 28):
        u = torch.rand(min(1 << 28, flat.numel() - b0), device=device, generator=g)
        quals[b0:b0 + u.numel()] = bins[torch.bucketize(u, cum)]
    return flat.contiguous(), offs.to(torch.int32), quals, lens


def make_reads(words, n, n_reads, M, device, seed, chunk=1_000_000, family=None):
    """150 bp reads drawn from the reference: 1 % substitutions, 0.1 %/base 1-3 bp indels,
    50 % reverse-complemented.  Returns (reads uint8 [R,M], truth pos, rc)."""
    import torch
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    out, pos_all, rc_all = [], [], []
    for b in range(0, n_reads, chunk):
        r = min(chunk, n_reads - b)
        pos = torch.randint(0, n - M - 8, (r,), device=device, generator=g, dtype=torch.int64)
        if family is not None:
            # 5 % of the reads start inside a copy of the repeat family (300 bp element: offsets 0..149 keep the read inside it)
            inside = torch.rand(r, device=device, generator=g) < 0.05
            pick = family[torch.randint(0, family.numel(), (r,), device=device, generator=g)]
            pos = torch.where(inside, pick + torch.randint(0, 150, (r,), device=device, generator=g), pos)
        j = torch.arange(M, device=device, dtype=torch.int64)[None, :]
        # indel: with probability 1-(1-0.001)^150 a read carries one indel of 1-3 bp
        has = torch.rand(r, device=device, generator=g) < (1.0 - (1.0 - 0.001) ** M)
        ip = torch.randint(5, M - 5, (r,), device=device, generator=g)[:, None]
        ig = torch.randint(1, 4, (r,), device=device, generator=g)[:, None]
        is_del = (torch.rand(r, device=device, generator=g) < 0.5)[:, None]
        hasc = has[:, None]
        src = torch.where(hasc & is_del & (j >= ip), j + ig, j)                     # deletion from the read
        src = torch.where(hasc & ~is_del & (j >= ip + ig), j - ig, src)             # insertion into the read
        sym = genome_symbols(words, pos[:, None] + src)
        ins = hasc & ~is_del & (j >= ip) & (j < ip + ig)
        rnd = torch.randint(0, 4, (r, M), device=device, generator=g, dtype=torch.uint8)
        sym = torch.where(ins, rnd, sym)
        sub = torch.rand(r, M, device=device, generator=g) < 0.01
        sym = torch.where(sub, (sym + 1 + rnd % 3) % 4, sym)                         # a different base
        rc = torch.rand(r, device=device, generator=g) < 0.5
        sym = torch.where(rc[:, None], 3 - sym.flip(1), sym)
        out.append(sym.contiguous()); pos_all.append(pos); rc_all.append(rc)
    return torch.cat(out), torch.cat(pos_all), torch.cat(rc_all)



def timed_ms(torch, fn, reps=5, warm=1):
    """median of `reps` event-timed calls on torch's current stream (which is the stream every library call is handed)"""
    for _ in range(warm):
        fn()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, b in ev:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    t = sorted(a.elapsed_time(b) for a, b in ev)
    return t[len(t) // 2]


def i32(torch, t):
    return torch.where(t >= 2 ** 31, t - 2 ** 32, t).to(torch.int32)


def run_configs(torch, amd, pipeline, fmi, genome, n, device, M, scale=1.0):
    """BASELINE.json configs 2, 4 (one GPU's slice) and 5 (1 M pairs) at kernel / composition level on the index the run already holds;
    each takes a few ms.  Not part of the timed step."""
    out = {}
    g = torch.Generator(device=device); g.manual_seed(2)
    # ---- config 2: 1 M x 22 bp seeds (90 % substrings of the text, 10 % random) through match() -- plain table + rank steps, and the
    #      reference's algorithm without a table (SURVEY 8d's unit of work: 32 B per distinct bwt_occ record + query + result) ----
    Q, L = int(1_000_000 * scale), 22
    starts = torch.randint(0, n - L, (Q,), device=device, generator=g, dtype=torch.int64)
    sym = genome_symbols(genome, starts[:, None] + torch.arange(L, device=device)[None, :])
    rnd = torch.randint(0, 4, (Q, L), device=device, generator=g, dtype=torch.uint8)
    sym[::10] = rnd[::10]
    padded = torch.zeros((Q, 24), dtype=torch.uint8, device=device); padded[:, :L] = sym
    qs = amd.PackedStringSet(pack4(padded.reshape(-1)), 4, Q, fixed_len=L, stride=24, device=device)
    _, blk = fmi.match(qs, amd.FM_NO_KMER_TABLE, want_blocks=True)
    alg = int((blk.to(torch.int64) & 0xFFFFFFFF).sum()) * 32 + Q * (11 + 8)
    ms_t = timed_ms(torch, lambda: fmi.match(qs, 0), reps=11, warm=2)
    ms_n = timed_ms(torch, lambda: fmi.match(qs, amd.FM_NO_KMER_TABLE), reps=11, warm=2)
    flt = amd.FMIndexFilter()
    total = flt.rank(fmi, qs)
    ms_l = timed_ms(torch, lambda: flt.locate(0, total), reps=11, warm=2) if total else None
    out["fm_seeds_1M"] = {"queries": Q, "seed_len": L, "ms": ms_t, "queries_per_s": Q / (ms_t * 1e-3),
                          "kernel": "fm_match_kernel<4,false,true,false> (plain k-mer table, then one rank step per symbol)",
                          # SURVEY 8d's algorithmic bytes are those of the REFERENCE's search (no table): they are a roofline fraction for the
                          # no-table kernel only; the table kernel skips 16 of the 22 steps, so it is quoted against them as a ratio, not a fraction
                          "alg_bytes": alg, "vs_reference_algorithm_at_peak": alg / (ms_t * 1e-3) / 1e9 / HBM_PEAK_GBS,
                          "no_table": {"kernel": "fm_match_kernel<4,false,false,false>: the reference's algorithm, one rank step per symbol", "ms": ms_n,
                                       "queries_per_s": Q / (ms_n * 1e-3), "alg_frac_of_hbm_peak": alg / (ms_n * 1e-3) / 1e9 / HBM_PEAK_GBS},
                          "locate_every_hit": {"hits": int(total), "ms": ms_l}}
    del qs, blk, padded, sym, rnd, starts, flt
    # ---- config 0's shape (sw-benchmark: 100 k x 100 bp patterns against ONE 4,096-symbol text, GLOBAL Gotoh(2,-1,-2,-1), text blocking): the config
    #      itself is the reference's CPU plumbing (scripts/bench_config1.py times that leg beside it); here the GPU side -- 100 k jobs are too few for
    #      a lane per job (1.5 waves per SIMD), the cooperative kernel gives a job four lanes -- checked against the oracle on a sample ----
    J, PM, TN = int(100_000 * scale), 100, 4096
    pats = torch.randint(0, 4, (J * PM,), device=device, generator=g, dtype=torch.uint8)
    twords = torch.randint(-2 ** 31, 2 ** 31 - 1, (TN // 16 + 8,), dtype=torch.int64, device=device, generator=g).to(torch.int32)   # 2-bit packed, random
    poffs = (torch.arange(J + 1, device=device) * PM).to(torch.int32)
    bj = amd.AlignmentBatch(pack4(pats), 4, poffs, twords, 2, torch.zeros(J, dtype=torch.int32, device=device),
                            torch.full((J,), TN, dtype=torch.int32, device=device), device=device)
    op = amd.BatchedAlignmentScore(amd.make_gotoh_aligner(amd.GLOBAL, amd.SimpleGotohScheme(2, -1, -2, -1)), text_blocking=True)
    ms0 = timed_ms(torch, lambda: op.enact(bj, PM, TN), reps=5, warm=1)
    sc0, _ = op.enact(bj, PM, TN)
    try:
        import numpy as np
        import oracle
        O = oracle.Oracle()
        smp = min(J, 256)
        pn = pats[:smp * PM].cpu().numpy(); tn_ = genome_symbols(twords, torch.arange(TN, device=device)).cpu().numpy()
        want, _ = O.full_gotoh_batch(oracle.GLOBAL, 1, oracle.Scheme(2, 1, 1, -2, -1, -2, -1), pn, (np.arange(smp + 1) * PM).astype(np.uint32),
                                     np.tile(tn_, smp), (np.arange(smp + 1) * TN).astype(np.uint32))
        equal = bool(np.array_equal(sc0[:smp].cpu().numpy(), want))
    except Exception as e:                                            # (the oracle is the checker only)
        equal = repr(e)
    out["sw_benchmark_100k"] = {"jobs": J, "pattern_len": PM, "text_len": TN, "ms": ms0, "gcups": J * PM * TN / (ms0 * 1e-3) / 1e9,
                                "kernel": "full_gotoh_coop_kernel<GLOBAL,4,25> (four lanes per job, boundary column in registers)",
                                "scores_equal_oracle_sample": equal}
    del pats, twords, poffs, bj
    # ---- config 4, one GPU's slice: 6.25 M pairs of (150, 181), band 31, LOCAL Gotoh match 2 / mismatch -6 (q >= 40) / gaps -8 -3 ----
    P = int(6_250_000 * scale)
    reads_sym, pos, rc = make_reads(genome, n, P, M, device, seed=4)
    reads_sym = torch.where(rc[:, None], 3 - reads_sym.flip(1), reads_sym)          # back on the forward strand
    r4 = pack4(reads_sym.reshape(-1))
    roffs = (torch.arange(P + 1, device=device) * M).to(torch.int32)
    wbeg = torch.clamp(pos - 15, min=0); wend = torch.clamp(wbeg + 31 + M, max=n)
    b = amd.AlignmentBatch(r4, 4, roffs, genome, 2, i32(torch, wbeg), i32(torch, wend), max_read_len=M, device=device)
    al = amd.make_gotoh_aligner(amd.LOCAL, amd.GotohScheme(2, 6, 6, -8, -3, -8, -3))
    ms = timed_ms(torch, lambda: amd.batch_banded_alignment_score(31, al, b), reps=5, warm=1)
    gc = P * 31 * M / (ms * 1e-3) / 1e9
    peak = 64 * 2 * 31 / (DP_ROW_INSTRUCTIONS * 4.0) * 1024 * GPU_CLOCK_GHZ
    out["banded_local_6.25M"] = {"pairs": P, "ms": ms, "gcups": gc, "pairs_per_s": P / (ms * 1e-3),
                                 "kernel": "banded_gotoh_band31_pk_kernel<LOCAL,4> (two alignments per lane, int16 packed)",
                                 "dp_issue_frac": gc / peak,
                                 "note": "dp_issue_frac: against the end-to-end row loop's issue bound (%d instructions per row); the LOCAL row also clamps at 0 and keeps a per-cell sink key" % DP_ROW_INSTRUCTIONS}
    del reads_sym, r4, b, wbeg, wend, pos, rc
    # ---- config 5 shape on one GPU: 1 M pairs, 2 x 150 bp, FR, insert N(350, 50), anchor banded + opposite mate full matrix, CIGARs of both ----
    Pp = int(1_000_000 * scale)
    g.manual_seed(77)
    ins = torch.clamp((torch.randn(Pp, device=device, generator=g) * 50 + 350).round().to(torch.int64), 160, 500)
    left = torch.randint(0, n - 520, (Pp,), device=device, generator=g, dtype=torch.int64)
    j = torch.arange(M, device=device, dtype=torch.int64)[None, :]

    def mate(posv, seed):
        gg = torch.Generator(device=device); gg.manual_seed(seed)
        symv = genome_symbols(genome, posv[:, None] + j)
        rndv = torch.randint(0, 4, (Pp, M), device=device, generator=gg, dtype=torch.uint8)
        sub = torch.rand(Pp, M, device=device, generator=gg) < 0.01
        return torch.where(sub, (symv + 1 + rndv % 3) % 4, symv)

    m1 = mate(left, 1); m2 = 3 - mate(left + ins - M, 2).flip(1)
    swap = torch.rand(Pp, device=device, generator=g) < 0.5
    b1 = pipeline.ReadBatch(pack4(torch.where(swap[:, None], m2, m1).reshape(-1)), Pp, M)
    b2 = pipeline.ReadBatch(pack4(torch.where(swap[:, None], m1, m2).reshape(-1)), Pp, M)
    pparams = pipeline.SeedExtendParams.end_to_end()
    pipeline.paired_end(fmi, genome, n, b1, b2, pparams, cigar_stride=16)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    po = pipeline.paired_end(fmi, genome, n, b1, b2, pparams, cigar_stride=16)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    paired = po["anchor"] >= 0
    conc = paired & (po["rc1"] != po["rc2"]) & ((po["pos1"] - po["pos2"]).abs() <= 500)
    # nvBowtie's own paired-end loop (Aligner::best_approx, paired form) over the same pairs: the C++ host loop, anchors iterated with the
    # pair-tightened threshold, the opposite mate scored for every anchor that passes (scores only: no traceback in this leg)
    loop = None
    try:
        s1 = pipeline.ReadBatch(pack4(torch.where(swap[:, None], m2, m1).flip(1).reshape(-1)), Pp, M)       # nvBowtie stores reads reversed
        s2 = pipeline.ReadBatch(pack4(torch.where(swap[:, None], m1, m2).flip(1).reshape(-1)), Pp, M)
        pipeline.nvbowtie_best_approx_paired_host(fmi, genome, n, s1, s2, pparams)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        lp = pipeline.nvbowtie_best_approx_paired_host(fmi, genome, n, s1, s2, pparams)
        torch.cuda.synchronize(); dl = time.perf_counter() - t0
        a1 = lp["best_a"][:, 0]; o1 = lp["best_o"][:, 0]
        is_pair = ((a1[:, 3] >> 2) & 1) == 1
        loop = {"ms": dl * 1e3, "pairs_per_s": Pp / dl, "paired_fraction": float(is_pair.float().mean()), "n_extensions": lp["n_extensions"],
                "n_opposite_alignments": lp["n_opposite"], "passes": lp["passes"],
                "best_pair_score_equals_composition": float(((a1[:, 0] + o1[:, 0]).to(torch.int64)[is_pair & paired] ==
                                                             (po["score1"] + po["score2"]).to(torch.int64)[is_pair & paired]).float().mean()),
                "host": "nvbio_host_best_approx_paired (lib/libnvbio_amd_host.so): aligner_best_approx_paired.h:84-200,590-1000"}
        del s1, s2, lp
    except Exception as e:
        loop = {"error": repr(e)}
    out["paired_end_1M"] = {"pairs": Pp, "ms": dt * 1e3, "nvbowtie_loop": loop, "pairs_per_s": Pp / dt, "paired_fraction": float(paired.float().mean()),
                            "concordant_fraction": float(conc.float().mean()), "with_cigars_of_both_mates": True,
                            "composition": "pipeline.paired_end: each mate anchored in turn (seed + banded extend), the other by full-matrix DP in nvBowtie's "
                                           "opposite-mate window, best pair, both mates traced back"}
    return out


def run_robust(torch, np, amd, pipeline, args, genome, n, R, M, device, rank, headline_ms):
    """The step on input that is NOT the friendliest instance of the workload: a repeat family in the reference (10^4 copies of a 300 bp
    element, 5 % of the reads drawn from inside a copy, seed-hit cap 16), per-base qualities under nvBowtie's default ramp (mismatch 2..6)
    and reads of different lengths (100..150, each seeded at its own interval and held to its own min score).  Rebuilds the index (the
    headline's handle has been closed: two 188 GB handles do not fit).  Results are checked against the plain operators."""
    t0 = time.time()
    family = plant_family(genome, n, args.robust_copies, device, seed=4321)
    fmi = amd.FMIndex.build(genome, n, kmer_len=args.kmer, sa_int=1,
                            table_flags=amd.FM_TABLE_CANONICAL if args.no_wide_table else amd.FM_TABLE_CANONICAL_WIDE)
    torch.cuda.synchronize()
    build_s = time.time() - t0
    reads_sym, truth_pos, truth_rc = make_reads(genome, n, R, M, device, seed=5000 + rank, family=family)
    min_len = min(100, M)
    flat, offs, quals, lens = make_ragged(reads_sym, min_len, device, seed=6000 + rank)
    del reads_sym
    reads4 = pack4(flat)
    del flat
    batch = pipeline.ReadBatch(reads4, R, M, quals=quals, offsets=offs)
    params = pipeline.SeedExtendParams.end_to_end(constant_quality=False, max_seed_hits=16)
    params.mapq = not args.no_mapq
    params.defer_heavy = not args.no_defer_heavy
    extras = {}

    def steps(k, timers=None):
        out = None
        pre = pipeline.seed_pass_begin(fmi, batch, params, 0, timers) if (not args.no_step_pipelining and k) else None
        for i in range(k):
            nxt = pipeline.seed_pass_begin(fmi, batch, params, (i + 1) & 1, timers) if (not args.no_step_pipelining and i + 1 < k) else None
            out = pipeline.seed_and_extend(fmi, genome, n, batch, params, timers, extras=extras, pre=pre)
            pre = nxt
        return out

    steps(1)
    timers = {}
    torch.cuda.synchronize(); w0 = time.perf_counter()
    K = max(2, min(args.steps, 3))
    bs, bp, brc, nc = steps(K, timers)
    torch.cuda.synchronize(); dt = (time.perf_counter() - w0) / K
    stage = {k: float(np.mean(event_ms(v))) for k, v in timers.items()}
    min_scores = extras.get("min_scores")
    aligned = bs >= min_scores if min_scores is not None else bs >= params.min_score_for(M)
    end_truth = torch.where(truth_rc, truth_pos + M, truth_pos + lens)      # a reverse-complemented read keeps the END of its locus when cut
    near = (bp - end_truth).abs() <= 40
    res = {"workload": ("3 Gbp i.i.d. reference with %d copies of a 300 bp element (5 %% of the reads drawn from inside a copy; max_seed_hits 16), "
                        "per-base qualities from {2,12,23,37,40} under nvBowtie's default ramp (mismatch 2..6, scoring.h:73-92), read lengths uniform "
                        "in [%d, %d] (seed interval and min score per read, mapping_inl.h:507-529)" % (args.robust_copies, min_len, M)),
           "reads": R, "ms_per_step": dt * 1e3, "reads_per_s": R / dt, "x_headline_ms": (dt * 1e3 / headline_ms) if headline_ms else None,
           "stage_ms": stage, "candidates_per_step": int(nc), "aligned_fraction": float(aligned.float().mean()),
           "correct_locus_fraction": float((aligned & near & (brc.bool() == truth_rc)).float().mean()),
           "index_and_tables_s": build_s,
           "switches": {"defer_heavy_searches": bool(params.defer_heavy), "quality_aware_first_pass": True, "one_pass_per_lane_for_ragged_reads": True}}
    if not args.no_plain_ab:
        # the same batch through the plain operators: match() + locate() of every seed of either strand, the DP for every candidate
        params.direct = False
        params.algo_flags = amd.ALN_NO_UNGAPPED_SCORE
        pipeline.seed_and_extend(fmi, genome, n, batch, params, None)
        torch.cuda.synchronize(); p0 = time.perf_counter()
        pbs, pbp, pbrc, pnc = pipeline.seed_and_extend(fmi, genome, n, batch, params, None)
        torch.cuda.synchronize(); pdt = time.perf_counter() - p0
        res["plain_operators"] = {"ms_per_step": pdt * 1e3, "results_equal": bool(torch.equal(pbs, bs) and torch.equal(pbp, bp) and torch.equal(pbrc, brc))}
    fmi.close()
    return res


def host_cpu_share():
    """(logical CPUs this process may run on, CPUs its cgroup quota grants or None): OpenMP's default of one thread per logical CPU
    oversubscribes a container whose quota is smaller -- round 2's baseline ran 128 threads on such a box"""
    aff = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    quota = None
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            quota = float(q) / float(per)
    except Exception:
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read()); per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                quota = q / per
        except Exception:
            pass
    return aff, quota


# ---------------------------------------------------------------------------------------------
def event_ms(pairs):
    return [a.elapsed_time(b) for a, b in pairs]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--ref-len", type=float, default=3e9, help="reference length (symbols)")
    ap.add_argument("--reads", type=float, default=10e6, help="reads per GPU per step")
    ap.add_argument("--read-len", type=int, default=150)
    ap.add_argument("--kmer", type=int, default=17, help="k of the k-mer SA-range table (0 disables; 17 = 128 GiB, 16 = 32 GiB + the 32 GiB position table)")
    ap.add_argument("--sa-int", type=int, default=1, help="SA sampling interval of the index built for the run")
    ap.add_argument("--verify", action="store_true", help="build the index with the SA/ISA/text verification shortcut (needs --sa-int 1)")
    ap.add_argument("--mode", choices=("e2e", "local"), default="e2e",
                    help="nvBowtie scoring mode of the extension: end-to-end (default, SURVEY 8d config 3) or local")
    ap.add_argument("--cpu-sample", type=int, default=0, help="reads timed on the host cores (0: sized for --cpu-seconds)")
    ap.add_argument("--cpu-seconds", type=float, default=30.0, help="target CPU time of the baseline sample")
    ap.add_argument("--no-direct", action="store_true", help="plain match() + locate() instead of the fused direct-position seed pass")
    ap.add_argument("--no-fused-seeds", action="store_true", help="match_direct + scan + locate_diagonals + dedupe as separate operators instead of the one-kernel seed pass")
    ap.add_argument("--no-canonical", action="store_true", help="the 128 GiB direct table and one seed pass per strand instead of the 64 GiB canonical "
                    "table and one pass for both strands")
    ap.add_argument("--no-wide-table", action="store_true", help="the canonical table with 8-byte entries (64 GiB at k = 17; a k-mer with two occurrences "
                    "then takes the group gather) instead of 16-byte ones (128 GiB, two occurrences in line)")
    ap.add_argument("--no-plain-ab", action="store_true", help="skip the (untimed) run through the plain operators without the two exact shortcuts")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-port-baseline", action="store_true", help="do not also time the oracle's C restatement beside the reference build")
    ap.add_argument("--force-dist", action="store_true",
                    help="initialise torch.distributed (RCCL) and run the result gather even with one rank (rehearsal of the N>1 path)")
    ap.add_argument("--no-mapq", action="store_true", help="leave nvBowtie's second-best bookkeeping and the mapping quality out of the step")
    ap.add_argument("--no-traceback", action="store_true", help="skip the (untimed) traceback-stage measurement")
    ap.add_argument("--algo-flags", type=int, default=0, help="A/B: NVBIO_ALN_* flags OR-ed into every alignment batch of the timed step")
    ap.add_argument("--pk-three-waves", action="store_true", help="A/B: the packed band-31 DP kernel built for 3 waves per SIMD (168 VGPRs, prologue state spills) instead of 2")
    ap.add_argument("--with-traceback", action="store_true", help="put the traceback of every aligned read's best alignment (CIGARs) inside the timed step")
    ap.add_argument("--build-breakdown", action="store_true", help="build the index once more without tables to report index_build_s and table_build_s separately")
    ap.add_argument("--max-seed-hits", type=int, default=0, help="extend at most this many SA rows of a seed's range (0: all; needed with --repeat-family)")
    ap.add_argument("--scaling", choices=("weak", "strong"), default="weak",
                    help="weak: every rank maps --reads reads per step; strong: --reads reads per step in all, split over the ranks")
    ap.add_argument("--side-priority", type=int, default=0, help="priority of the seed pass's side stream (-1 = high)")
    ap.add_argument("--no-overlap-leg", action="store_true", help="skip the (extra) measurement of the step with the seed pass beside the extension")
    ap.add_argument("--overlap-grid-blocks", type=int, default=448, help="workgroups of the seed pass in that measurement (a multiple of 64)")
    ap.add_argument("--seed-grid-blocks", type=int, default=0, help="workgroups of the seed-pass launch (a multiple of 64; 0 = the library's 16384): with "
                    "--overlap-seed-pass a small grid leaves SIMD room for the extension's kernels")
    ap.add_argument("--overlap-seed-pass", action="store_true", help="the next batch's seed pass on a side stream BESIDE this batch's extension: 5.62 instead of 5.76 ms per step, "
                    "but each kernel then runs slower than alone (the seed pass 3.3 instead of 2.3 ms) and the roofline of the launch would be that of a shared GPU: off by default")
    ap.add_argument("--no-step-pipelining", action="store_true",
                    help="do not enqueue the next step's seed pass ahead of this step's extension (the host then waits for the seed pass's counts with the GPU idle)")
    ap.add_argument("--no-defer-heavy", action="store_true", help="the searches the canonical table cannot answer run inside the seed pass instead of as a dense launch behind it")
    ap.add_argument("--no-sweep", action="store_true", help="skip the (untimed) 1-GPU sweep of the step over R/8 .. R reads")
    ap.add_argument("--no-nvbowtie-mode", action="store_true", help="skip the (untimed) run of nvBowtie's own selection / effort loop (C++ host loop over the C ABI)")
    ap.add_argument("--no-cpp-host", action="store_true", help="skip the run of the same step through the C++ host program (lib/fmmap_amd) on the same reads")
    ap.add_argument("--no-configs", action="store_true", help="skip the (untimed) kernel-level runs of BASELINE configs 2, 4 and 5")
    ap.add_argument("--no-robust", action="store_true", help="skip the robust-input step (repeat family + per-base qualities + ragged reads: rebuilds the index)")
    ap.add_argument("--robust-copies", type=int, default=10000, help="copies of the 300 bp element planted for the robust-input step")
    ap.add_argument("--repeat-family", type=int, default=0, metavar="COPIES",
                    help="plant COPIES copies of a random 300 bp element in the reference (and draw 5 %% of the reads from them): a repeat-rich workload")
    args = ap.parse_args()

    import numpy as np
    import torch
    import __graft_entry__ as ge
    amd = ge.load_package()
    pipeline = importlib.import_module("nvbio_gpl_amd.pipeline")
    sharding = importlib.import_module("nvbio_gpl_amd.sharding")

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    device = "cuda:%d" % local_rank
    dist = None
    if world > 1 or args.force_dist:
        import torch.distributed as dist_mod
        dist = dist_mod
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        dist.init_process_group(backend="nccl", rank=rank, world_size=world, device_id=torch.device(device))

    n = int(args.ref_len)
    R = int(args.reads)
    if args.scaling == "strong":
        R = R // world                                            # strong scaling: the batch of --reads reads is split over the ranks
    M = args.read_len
    t0 = time.time()
    genome = make_reference(n, device, seed=1234)                 # every rank holds the same replica
    family = None
    if args.repeat_family:
        family = plant_family(genome, n, args.repeat_family, device, seed=4321)
    torch.cuda.synchronize()
    t1 = time.time()
    # the suffix sort / BWT / SA alone (no tables), timed on its own so that the table build can be reported separately
    index_only_s = None
    if args.build_breakdown:
        tb0 = time.time()
        tmp = amd.FMIndex.build(genome, n, kmer_len=0, sa_int=args.sa_int)
        torch.cuda.synchronize()
        index_only_s = time.time() - tb0
        tmp.close(); del tmp
        torch.cuda.empty_cache()
        t1 = time.time()
    canonical = (not args.no_canonical and args.sa_int == 1 and args.kmer >= 15 and args.kmer % 2 == 1 and not args.no_direct
                 and not args.no_fused_seeds)               # the canonical table serves seeds of k .. k + 7 symbols (22-mers: k >= 15)
    try:
        fmi = amd.FMIndex.build(genome, n, kmer_len=args.kmer, sa_int=args.sa_int, verify=(args.sa_int == 1 and args.verify),
                                table_flags=(amd.FM_TABLE_CANONICAL if args.no_wide_table else amd.FM_TABLE_CANONICAL_WIDE) if canonical else 0)
    except amd.NvbioError as e:
        if args.kmer <= 16 or "memory" not in str(e):
            raise
        # the k = 17 table needs 128 + 32 GiB while it is built: fall back to k = 16 on a card that cannot give that
        log("k-mer table k=%d does not fit (%s): falling back to k=16" % (args.kmer, e))
        args.kmer = 16
        torch.cuda.empty_cache()
        fmi = amd.FMIndex.build(genome, n, kmer_len=16, sa_int=args.sa_int, verify=(args.sa_int == 1 and args.verify))
    torch.cuda.synchronize()
    t2 = time.time()
    log("reference %d symbols generated in %.2fs, index built on the GPU in %.2fs (k-mer table k=%d, %.2f GB owned)"
        % (n, t1 - t0, t2 - t1, args.kmer, fmi.device_bytes() / 1e9))
    build_s = t2 - t1
    reads_sym, truth_pos, truth_rc = make_reads(genome, n, R, M, device, seed=1000 + rank, family=family)
    reads4 = pack4(reads_sym.view(-1))
    torch.cuda.synchronize()
    log("%d reads x %d bp per rank generated in %.2fs" % (R, M, time.time() - t2))
    batch = pipeline.ReadBatch(reads4, R, M)
    params = pipeline.SeedExtendParams.end_to_end() if args.mode == "e2e" else pipeline.SeedExtendParams()
    params.direct = not args.no_direct
    params.max_seed_hits = args.max_seed_hits or None
    if args.pk_three_waves:
        params.algo_flags = amd.ALN_PK_THREE_WAVES
    if args.algo_flags:
        params.algo_flags = (params.algo_flags or 0) | args.algo_flags
    params.fused_seed_pass = not args.no_fused_seeds
    params.seed_grid_blocks = args.seed_grid_blocks
    params.defer_heavy = not args.no_defer_heavy
    params.mapq = not args.no_mapq               # score_reduce's second-best alignment + BowtieMapq2, inside the timed step
    sv = params.scheme.c
    scheme_t = tuple(int(getattr(sv, f)) for f, _ in sv._fields_)
    min_score = params.min_score_for(M)

    # the only exchange of the path: every rank's per-read best (score, position, strand), 8 bytes per read, to
    # rank 0 over RCCL -- enqueued asynchronously after the batch's kernels so that it overlaps the next batch
    gatherer = sharding.ResultGatherer(dist, world, rank, R, device, dst=0) if dist is not None else None

    extras = {}

    def step(timers=None, pre=None):
        if args.with_traceback:
            bs, bp, brc, nc, bwb, _ = pipeline.seed_and_extend(fmi, genome, n, batch, params, timers, extras=extras, return_windows=True, pre=pre)
            extras["traceback"] = pipeline.traceback_best_all(genome, n, batch, params, extras["best_keys"], bwb, cigar_stride=TB_STRIDE, timers=timers)
        else:
            bs, bp, brc, nc = pipeline.seed_and_extend(fmi, genome, n, batch, params, timers, extras=extras, pre=pre)
        if gatherer is not None:
            gatherer.submit(sharding.pack_result64(bs, bp, brc))
        return bs, bp, brc, nc

    # Steps are software-pipelined as a caller that streams batches would: the seed pass of step i+1 is enqueued BEFORE the extension of
    # step i, so the host learns i+1's candidate count (the one host synchronisation of a step: it sizes the extension's launches) while the
    # GPU extends batch i and the GPU never waits for the host.  K steps still are K seed passes and K extensions inside the timed region.
    can_pipe = (not args.no_step_pipelining and params.direct and fmi.supports_direct() and params.fused_seed_pass and bool(fmi.canonical))

    side = torch.cuda.Stream(device=device, priority=args.side_priority) if (can_pipe and args.overlap_seed_pass) else None

    def run_steps(k, timers=None):
        # the seed pass of batch i+1 on a side stream beside the extension of batch i; a slot's buffers are rewritten only behind the extension that read them
        out = None
        done = [None, None]
        if side is not None:
            side.wait_stream(torch.cuda.current_stream(device))
        pre = pipeline.seed_pass_begin(fmi, batch, params, 0, timers, stream=side) if (can_pipe and k) else None
        for i in range(k):
            nxt = pipeline.seed_pass_begin(fmi, batch, params, (i + 1) & 1, timers, stream=side, after=done[(i + 1) & 1]) if (can_pipe and i + 1 < k) else None
            out = step(timers, pre)
            if side is not None:
                done[i & 1] = torch.cuda.Event(); done[i & 1].record()
            pre = nxt
        return out

    run_steps(args.warmup)
    if gatherer is not None:
        gatherer.wait()
    timers = {}
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    w0 = time.perf_counter()
    bs, bp, brc, nc = run_steps(args.steps, timers)
    if gatherer is not None:
        gatherer.wait()                                               # every batch's results have landed on rank 0
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    elapsed = time.perf_counter() - w0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # ---- quality of the result (not timed): reads aligned, and aligned where they came from ----
    aligned = bs >= min_score
    end_truth = truth_pos + M                                         # approximate end of the true locus
    near = (bp - end_truth).abs() <= 40
    frac_aligned = float(aligned.float().mean())
    frac_correct = float((aligned & near & (brc.bool() == truth_rc)).float().mean())

    # ---- traceback of every read's best alignment (not part of the timed step) ----
    tb_ms, tb_info = None, None
    if not args.no_traceback:
        tbx = {}
        _, _, _, _, bwb, _ = pipeline.seed_and_extend(fmi, genome, n, batch, params, None, return_windows=True, extras=tbx)
        tms = []
        for _ in range(3):
            tt = {}
            tsc, tpos, tsrc, tsnk, tcig, tln = pipeline.traceback_best_all(genome, n, batch, params, tbx["best_keys"], bwb, cigar_stride=TB_STRIDE, timers=tt)
            torch.cuda.synchronize()
            tms.append(event_ms(tt["traceback"])[0])
        tb_ms = float(np.median(tms))
        lens_all = tln.to(torch.int64) & 0xFFFFFFFF
        traced = lens_all > 0
        lens_i = lens_all[traced]
        tb_info = {"kernels": "traceback_best_batch_kernel (the batch of every read's best alignment, built on the device) + ungapped_traceback_kernel<31,%s,4,2> "
                              "(diagonal check, settles ungapped reads) + banded_gotoh_traceback_kernel "
                              "(the rest: one DP pass writing 16 B of direction vectors per row, walk back to a run-length CIGAR)%s"
                              % ("SEMI_GLOBAL" if args.mode == "e2e" else "LOCAL",
                                 "; score and sink handed over from the scoring pass" if args.mode == "e2e" else "; scoring pass re-run inside"),
                   "reads": int(traced.sum()), "ms": tb_ms, "reads_per_s": int(traced.sum()) / (tb_ms * 1e-3),
                   "gapped_fraction": float((lens_i > 1).float().mean()),
                   "scores_equal_scoring_pass": bool(torch.equal(tsc[traced], bs[traced]) and torch.equal(traced, aligned)),
                   "mean_cigar_elements": float(lens_i.float().mean()), "max_cigar_elements": int(lens_i.max()),
                   "cigar_stride": TB_STRIDE, "cigars_truncated": int((lens_i > TB_STRIDE).sum())}
        del bwb, tsc, tpos, tsrc, tsnk, tcig, tln, tbx

    # ---- stage times and the roofline of the dominant HBM kernel (the seed pass) --------------------------------
    stage_ms = {k: float(np.mean(event_ms(v))) for k, v in timers.items()}
    spr = (M - params.seed_len) // params.interval_for(M) + 1
    n_seeds = R * spr
    qs = amd.PackedStringSet(reads4, 4, n_seeds, fixed_len=params.seed_len, stride=M, device=device, seeds_per_string=spr,
                             seed_interval=params.interval_for(M))
    strands = ((0, 0), (1, amd.FM_SCAN_FORWARD | amd.FM_COMPLEMENT))
    use_direct = bool(params.direct and fmi.supports_direct())
    use_fused = bool(use_direct and params.fused_seed_pass)
    use_both = bool(use_fused and fmi.canonical)                 # one launch serves both strands
    launches = 1 if use_both else 2
    seed_tag = (SEED_BOTH_KERNEL_TAG % ("false" if args.no_wide_table else "true", "false" if args.no_defer_heavy else "true")) if use_both else SEED_KERNEL_TAG
    # (1) bytes the TIMED launch has to move, at the 64-byte sector granularity of the fabric (outside the timed region, by the
    #     kernel's accounting instantiation, NVBIO_FM_COUNT_SECTORS): every gather of a search -- direct-table entry, group of a
    #     2..7-occurrence k-mer, bwt_occ records of the rank steps that are left, SA word, text words -- counted as one sector
    #     per distinct 64 bytes, plus what the launch streams: the packed reads once, the tiles' keys and counts written, then
    #     read and written again by the compaction, the residual list.
    launch_bytes = sectors = streamed = None
    if use_both:
        bufs = fmi.match_seed_diagonals_both(qs, M, flags=amd.FM_COUNT_SECTORS)
        c = bufs["counts"].cpu().numpy().astype(np.int64) & 0xFFFFFFFF
        n_tiles = -(-R // (64 // spr))
        sectors = float(int(c[4] | (c[5] << 32)))
        streamed = float(R * M // 2 + 3 * 8 * int(c[0]) + 3 * 4 * n_tiles + 12 * int(c[1] + c[2]))
        launch_bytes = sectors * LINE + streamed
        del bufs
    elif use_fused:
        acc = []
        for strand, flags in strands:
            bufs = fmi.match_seed_diagonals(qs, flags | amd.FM_COUNT_SECTORS, M, strand)
            c = bufs["counts"].cpu().numpy().astype(np.int64) & 0xFFFFFFFF
            n_tiles = -(-R // (64 // spr if spr <= 64 else 1))
            sect = int(c[2] | (c[3] << 32))
            acc.append((sect, R * M // 2 + 3 * 8 * int(c[0]) + 3 * 4 * n_tiles + 12 * int(c[1])))
            del bufs
        sectors = float(np.mean([a_[0] for a_ in acc]))
        streamed = float(np.mean([a_[1] for a_ in acc]))
        launch_bytes = sectors * LINE + streamed
    # (2) algorithmic bytes of the REFERENCE's algorithm for the same launch: 32 B x distinct bwt_occ records its
    #     backward search touches (counted by the match kernel's NO_KMER_TABLE accounting mode) + 11 B of query symbols
    #     (22 x 4 bit) + 8 B of result per query -- and the time of that algorithm's own kernel (no table, every symbol stepped)
    blocks = 0
    for _, flags in strands:
        _, blk = fmi.match(qs, flags | amd.FM_NO_KMER_TABLE, want_blocks=True)
        blocks += int((blk.to(torch.int64) & 0xFFFFFFFF).sum())
        del blk
    alg_bytes_per_launch = (blocks * 32 + 2 * n_seeds * (11 + 8)) / float(launches)
    nt = []
    for _, flags in strands:
        a_ev, b_ev = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a_ev.record()
        fmi.match(qs, flags | amd.FM_NO_KMER_TABLE)
        b_ev.record()
        torch.cuda.synchronize()
        nt.append(a_ev.elapsed_time(b_ev))
    no_table_ms = float(np.sum(nt)) / launches                # per launch of the timed pass (both strands when it serves both)

    # ---- the same step through the plain operators (outside the timed region): match() + locate() without the fused
    #      direct-position seed pass, and the banded DP for every candidate without the ungapped shortcut.  Results are
    #      identical by construction (and by test); the stage times show what the two exact shortcuts buy. ----
    plain = {}
    if not args.no_plain_ab:
        prev_direct = params.direct
        params.direct = False
        params.algo_flags = amd.ALN_NO_UNGAPPED_SCORE | (amd.ALN_PK_THREE_WAVES if args.pk_three_waves else 0)   # algo_flags: DP for every candidate
        pipeline.seed_and_extend(fmi, genome, n, batch, params, None)                 # warm
        pt = {}
        torch.cuda.synchronize(); p0 = time.perf_counter()
        pbs, pbp, pbrc, pnc = pipeline.seed_and_extend(fmi, genome, n, batch, params, pt)
        torch.cuda.synchronize(); pdt = time.perf_counter() - p0
        params.algo_flags = (amd.ALN_PK_THREE_WAVES if args.pk_three_waves else 0) | (args.algo_flags or 0)
        params.direct = prev_direct
        pst = {k: float(np.mean(event_ms(v))) for k, v in pt.items()}
        plain = {"ms_per_step": pdt * 1e3, "match_ms_per_launch": 0.5 * (pst.get("match_fw", 0.0) + pst.get("match_rc", 0.0)),
                 "locate_ms_per_launch": pst.get("locate", 0.0), "extend_ms": pst.get("extend_fw", 0.0) + pst.get("extend_rc", 0.0) + pst.get("extend", 0.0),
                 "extend_gcups": float(pnc) * params.band * M / ((pst.get("extend_fw", 0.0) + pst.get("extend_rc", 0.0) + pst.get("extend", 0.0)) * 1e-3) / 1e9,
                 "kmer_table": (args.kmer - 1) if (fmi.supports_direct() and args.kmer >= 2) else args.kmer,   # the handle's plain table
                 "results_equal": bool(torch.equal(pbs, bs) and torch.equal(pbp, bp) and torch.equal(pbrc, brc))}
    match_ms = stage_ms.get("match_both", 0.0) if use_both else 0.5 * (stage_ms.get("match_fw", 0.0) + stage_ms.get("match_rc", 0.0))
    if launch_bytes is None:                    # the separate operators: what their match kernel moves is not accounted; use the reference's bytes
        launch_bytes = alg_bytes_per_launch
    achieved = launch_bytes / (match_ms * 1e-3) / 1e9 if match_ms > 0 else 0.0
    traffic = traffic_src = None
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(tpath):
        try:
            tj = json.load(open(tpath))
            if (tj.get("ref_len") == n and tj.get("reads") == R and tj.get("kmer") == args.kmer and tj.get("sa_int", 16) == args.sa_int
                    and bool(tj.get("direct", False)) == use_direct and bool(tj.get("fused", False)) == use_fused
                    and tj.get("kernel_tag") == seed_tag):
                traffic = tj.get("match_hbm_bytes_per_launch")
                traffic_src = "profiles/traffic.json (tag %s): %s; a profile of this configuration, not measured in this run" % (tj.get("tag"), tj.get("note"))
        except Exception:
            traffic = None
    # the measured ceiling for this access pattern: a chain of two dependent random 8-byte gathers over the table's footprint (128 GiB
    # direct table, 64 GiB canonical table)
    # (scripts/ubench/gather_rate.hip, profiles/r02_gather_rate.jsonl)
    # the chip's rate of random 128-byte lines over the table's footprint, from the round-3 micro-benchmark (scripts/ubench/gather2.hip ->
    # profiles/r03_gather2_sweep.jsonl: flat in occupancy, gathers in flight per lane and chain length; its counters: r03_gather2_pmc_128GiB.jsonl)
    ceiling = None
    gpath = os.path.join(ROOT, "profiles", "r03_gather2_sweep.jsonl")
    if os.path.exists(gpath):
        want_log = 36 if (use_both and args.no_wide_table) else 37
        best_fit = None
        for line in open(gpath):
            try:
                gj = json.loads(line)
            except Exception:
                continue
            if gj.get("elem_bytes") == 8 and gj.get("chain") == 1 and gj.get("inflight_per_lane") == 1 and gj.get("waves_per_simd") == 8 and not gj.get("precomputed_addresses"):
                if best_fit is None or abs(gj["footprint_log2"] - want_log) < abs(best_fit["footprint_log2"] - want_log):
                    best_fit = gj
        if best_fit is not None:
            ceiling = best_fit.get("G_gathers_per_s")
    cells = float(nc) * params.band * M
    extend_ms = stage_ms.get("extend_fw", 0.0) + stage_ms.get("extend_rc", 0.0) + stage_ms.get("extend", 0.0) + stage_ms.get("extend_res", 0.0)
    step_ms = elapsed / args.steps * 1e3
    build = {"index_and_tables_s": build_s}
    if index_only_s is not None:
        build.update(index_build_s=index_only_s, table_build_s=max(build_s - index_only_s, 0.0))
    build["break_even_reads"] = int(build_s / (step_ms * 1e-3) * R)      # reads mapped in the time the index took to build

    result = {
        "metric": "aligned reads/sec (150 bp single-end, 3 Gbp ref); GCUPS of the banded extend pass in `extend`",
        "value": world * R * args.steps / elapsed,
        "unit": "reads/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": step_ms,
        "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None,
        "dtype": "u32 (FM-index coordinates) / i16x2 (band-31 DP, two alignments per 32-bit lane: 16-bit integer scores, held in binary16 lanes where every score is an integer of magnitude <= 2040 -- exact -- and in int16 lanes otherwise; bound-checked on the host)",
        "data": "synthetic",
        "config": {"workload": ("nvbowtie-se-150bp-3gbp" if (n == 3_000_000_000 and R == 10_000_000 and M == 150 and not args.repeat_family) else "custom"),
                   "ref_len": n, "reads_per_gpu": R, "read_len": M, "seed_len": params.seed_len,
                   "seed_interval": params.interval_for(M), "seeds_per_read": 2 * spr, "band": params.band,
                   "alignment": ("end-to-end (SEMI_GLOBAL) Gotoh, match 0, mismatch -6 (constant q>=40), gaps -8/-3, min score -0.6-0.6L"
                                 if args.mode == "e2e" else "local Gotoh, match 2, mismatch -2 (no qualities), gaps -8/-3, min score 10 ln L"), "kmer_table": args.kmer, "canonical_table": use_both, "wide_entries": bool(use_both and not args.no_wide_table), "sa_int": args.sa_int, "match_direct": use_direct, "fused_seed_pass": use_fused, "sa_isa_verify": bool(args.sa_int == 1 and args.verify),
                   "index_bytes_per_gpu": fmi.device_bytes(), "index_bytes_per_reference_base": fmi.device_bytes() / float(n),
                   "parallelism": "read-shard x%d" % world, "reads_per_step_all_ranks": world * R,
                   "steps_pipelined": bool(can_pipe), "seed_pass_beside_extension": bool(can_pipe and args.overlap_seed_pass), "defer_heavy_searches": bool(params.defer_heavy),
                   "traceback_in_step": bool(args.with_traceback), "repeat_family_copies": args.repeat_family,
                   "max_seed_hits": params.max_seed_hits},
        "left_out_of_the_step": {"note": "the timed step re-runs one HBM-resident batch: no H2D/D2H, no index build"
                                         + ("" if args.with_traceback else ", no traceback (see `traceback`; --with-traceback puts it inside)"),
                                 "build": build},
        "aligned_fraction": frac_aligned, "correct_locus_fraction": frac_correct,
        "stage_ms": stage_ms,
        "roofline": {"kernel": (("seed pass launch of BOTH strands, %d seed windows = %d searches: %s (match + locate + diagonal key + adjacent dedupe of every seed "
                                 "and of its reverse complement from one canonical-table gather per window, one wave per tile of whole reads) followed by the "
                                 "tile-count scan and fm_seed_compact_kernel (a few %% of the launch)" % (n_seeds, 2 * n_seeds, seed_tag)) if use_both else
                                ("seed pass launch of one strand, %d seeds: %s (match + locate + diagonal key + adjacent dedupe of every seed, one wave per "
                                 "tile of whole reads) followed by the tile-count scan and fm_seed_compact_kernel (a few %% of the launch)" % (n_seeds, SEED_KERNEL_TAG)) if use_fused else
                                ("fm_match_kernel<4,false,true,%s> (seed pass, one strand of %d seeds per launch%s)"
                                 % ("true" if use_direct else "false", n_seeds,
                                    "; single-row searches finish on the text and return positions: match + locate fused" if use_direct else ""))),
                     "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS,
                     "bytes": ("bytes the timed launch has to move at the granularity the hardware moves them: every gather is ONE 128-byte fabric read "
                               "whatever its width, the memory kind or the cache-policy bits (TCC_EA0_RDREQ_128B = gathers, _64B = _32B = 0: "
                               "profiles/r03_gather2_modes.jsonl), so gathers (counted by the kernel's accounting instantiation, outside the timed "
                               "region) x 128 + what the launch streams (packed reads, keys / counts written and compacted)"),
                     "bytes_per_launch": launch_bytes, "gathered_sectors_per_launch": sectors, "ms_per_launch": match_ms,
                     "sectors_per_seed": (sectors / n_seeds) if sectors else None,
                     # of those bytes, the ones the searches look at: the entry / group / record a gather was issued for (<= 64 B of each line)
                     "useful_bytes_per_launch": (sectors * SECTOR + streamed) if (sectors and streamed is not None) else None,
                     "useful_frac": ((sectors * SECTOR + streamed) / (match_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if (sectors and streamed is not None and match_ms > 0) else None,
                     "frac_of_measured_copy_rate": (achieved / 6290.0) if achieved else None,     # MI355X_MICROARCH.md: 6.29 TB/s float4 copy
                     "traffic": traffic, "traffic_source": traffic_src,
                     "traffic_frac": (traffic / (match_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if (traffic and match_ms > 0) else None,
                     "launches_per_step": launches,
                     "queries_per_s": n_seeds * (2 if use_both else 1) / (match_ms * 1e-3) if match_ms > 0 else 0.0,
                     "gather_rate_G_per_s": (sectors / (match_ms * 1e-3) / 1e9) if (sectors and match_ms > 0) else None,
                     "gather_ceiling_G_per_s": ceiling,
                     "frac_of_gather_ceiling": (sectors / (match_ms * 1e-3) / 1e9 / ceiling) if (sectors and ceiling and match_ms > 0) else None,
                     # not a roofline fraction: how the timed launch compares with the reference's algorithm running at the HBM peak
                     "vs_reference_algorithm_at_peak": (alg_bytes_per_launch / (match_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if match_ms > 0 else None,
                     "reference_algorithm": {"kernel": "fm_match_kernel<4,false,false,false>: match() as the reference runs it, one rank step per symbol, no table",
                                             "algorithmic_bytes_per_launch": alg_bytes_per_launch, "ms_per_launch": no_table_ms,
                                             "achieved": alg_bytes_per_launch / (no_table_ms * 1e-3) / 1e9,
                                             "frac": alg_bytes_per_launch / (no_table_ms * 1e-3) / 1e9 / HBM_PEAK_GBS}},
        "extend": {"kernel": ("ungapped_e2e31_kernel<4> (31 diagonals by XOR + popcount on bit planes; settles every candidate whose best diagonal "
                              "beats any gapped alignment; second / third chance and the gap chance over dense lists) + banded_gotoh_band31_pk_kernel<SEMI_GLOBAL,4> over the rest (two alignments per lane, "
                              "binary16 packed: exact for these scores, v_pk_maximum3_f16)") if args.mode == "e2e" else
                             "banded_gotoh_band31_pk_kernel<LOCAL,4> (two alignments per lane, int16 packed)",
                   "bound": "valu (packed 16-bit lanes; MFMA not applicable)",
                   "candidates_per_step": int(nc), "cells_per_step": cells, "ms": extend_ms,
                   # GCUPS of the DP kernel itself: the A/B run below, where the DP computes every cell of every candidate
                   "gcups": plain.get("extend_gcups") if plain else None,
                   # cells of the full band DP / time of the step's extension stage, in which the exact shortcuts settle most candidates
                   "effective_gcups": cells / (extend_ms * 1e-3) / 1e9 if extend_ms > 0 else 0.0,
                   # what bounds the DP kernel: VALU issue.  One row of a wave = 64 lanes x 2 alignments x 31 cells in DP_ROW_INSTRUCTIONS
                   # instructions (the row loop's listing: profiles/r03_pk_row_loop_f16.s), 4 cycles each on a 16-lane SIMD, 1024 SIMDs
                   "dp_issue_bound": (lambda peak: {"instructions_per_row": DP_ROW_INSTRUCTIONS, "cells_per_wave_row": 64 * 2 * 31,
                                                     "cycles_per_instruction": 4, "simds": 1024, "clock_ghz": GPU_CLOCK_GHZ, "peak_gcups": peak,
                                                     "frac": (plain.get("extend_gcups") / peak) if plain and plain.get("extend_gcups") else None})(
                                          64 * 2 * 31 / (DP_ROW_INSTRUCTIONS * 4.0) * 1024 * GPU_CLOCK_GHZ)},
        "traceback": tb_info,
        "mapq": ({"evaluator": "BowtieMapq2 over (best, second best) of nvBowtie's score_reduce; inside the timed step (stage_ms.mapq)",
                  "second_alignment_fraction": float((extras["second"] != 0).float().mean()),
                  "mean": float(extras["mapq"].float().mean()), "fraction_ge_30": float((extras["mapq"] >= 30).float().mean())}
                 if params.mapq and "mapq" in extras else None),
        "plain_operators": plain or None,
    }

    # ---- CPU baseline on a bounded sample of the same reads, all host cores ------------------------
    # kind "reference": the reference's own host templates (oracle/_ref, compiled from /root/reference in
    #   the development container; the .so travels with the snapshot) -- match() / locate() /
    #   banded_alignment_score<31> under OpenMP, run on the index the GPU built;
    # kind "port": the oracle's C restatement, when oracle/_ref is not there.
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        import oracle
        from oracle import cpu_pipeline
        O = oracle.Oracle()
        b_occ, ssa = fmi.arrays()
        v = fmi.view()
        # both CPU paths use the reference's K = 16 sampled SA: take every (16/sa_int)-th entry
        assert 16 % args.sa_int == 0, "the CPU baseline needs sa_int <= 16"
        hidx = oracle.HostIndex(n, v.primary, [v.L2[i] for i in range(5)], amd.u32(b_occ),
                                np.ascontiguousarray(amd.u32(ssa)[::16 // args.sa_int]))
        del b_occ, ssa
        g_host = amd.u32(genome)
        use_ref = oracle.Reference.available()
        aff, quota = host_cpu_share()
        cores_granted = max(1, min(aff, int(quota)) if quota else aff)
        if use_ref:
            Rf = oracle.Reference()
            Rf.adopt_index(hidx)

            def run(reads_np):
                tm = {}
                out = cpu_pipeline.seed_and_extend_ref(Rf, O, hidx, g_host, n, reads_np, timing=tm, aln_type=params.aln_type,
                                                       scheme=oracle.Scheme(*scheme_t))
                return out, tm["ref_seconds"]
            Rf.set_num_threads(cores_granted)
            cores = Rf.num_threads()
        else:
            O.set_num_threads(cores_granted)
            def run(reads_np):
                c0 = time.perf_counter()
                out = cpu_pipeline.seed_and_extend_cpu(O, hidx, g_host, n, reads_np, genome_is_packed=True,
                                                       aln_type=params.aln_type, scheme=oracle.Scheme(*scheme_t))
                return out, time.perf_counter() - c0
            cores = O.num_threads()
        probe = reads_sym[:20000].cpu().numpy()
        run(probe)                                                   # page the index in
        _, pdt = run(probe)                                          # estimate the rate
        Rs = args.cpu_sample or int(len(probe) / pdt * args.cpu_seconds)
        Rs = max(1000, min(Rs, R, 2_000_000))
        (cs, cp, crc, cnc), cdt = run(reads_sym[:Rs].cpu().numpy())
        same = bool(np.array_equal(cs, bs[:Rs].cpu().numpy()) and np.array_equal(cp, bp[:Rs].cpu().numpy())
                    and np.array_equal(crc, brc[:Rs].cpu().numpy()))
        result["cpu_baseline"] = {
            "value": Rs / cdt, "unit": "reads/s", "cores": cores, "kind": "reference" if use_ref else "port",
            "sample": "first %d reads of rank 0's batch; whole path (2x9 exact 22-mer seeds, locate of every hit, band-31 "
                      "Gotoh of every candidate, same mode) through %s, OpenMP over work items; %.1f s"
                      % (Rs, "the reference's own host templates (oracle/_ref)" if use_ref else "the oracle's C restatement", cdt),
            "results_equal_gpu": same,
            "logical_cpus": aff, "cgroup_quota_cpus": quota,
            "threads_note": ("threads = CPUs the process is granted (min of its affinity mask and its cgroup quota); round 2 ran one thread per logical CPU "
                             "(128 / 256 by box) whatever the quota")}
        log("cpu baseline (%s): %d reads in %.1fs on %d threads (equal to GPU results: %s)"
            % (result["cpu_baseline"]["kind"], Rs, cdt, cores, same))
        if use_ref and not args.no_port_baseline:
            # the oracle's C restatement on (a tenth of) the same sample, beside the reference build
            O.set_num_threads(cores)
            Rp = max(1000, Rs // 10)
            sub_np = reads_sym[:Rp].cpu().numpy()
            c0 = time.perf_counter()
            pout = cpu_pipeline.seed_and_extend_cpu(O, hidx, g_host, n, sub_np, genome_is_packed=True,
                                                    aln_type=params.aln_type, scheme=oracle.Scheme(*scheme_t))
            pdt_cpu = time.perf_counter() - c0
            result["cpu_baseline"]["port"] = {"value": Rp / pdt_cpu, "unit": "reads/s", "cores": cores, "kind": "port", "reads": Rp,
                                              "note": "the oracle's C restatement; includes its numpy glue (the reference leg counts the time inside the reference's functions only)",
                                              "results_equal_reference": bool(np.array_equal(pout[0], cs[:Rp]) and np.array_equal(pout[1], cp[:Rp]))}

    # ---- strong-scaling readiness on ONE GPU: the same step over R/8, R/4, R/2 reads (what each of 8 / 4 / 2 ranks would map of a batch of R
    #      reads split over them): fixed costs -- launches, the step's host synchronisation -- show as ms/step that does not shrink with R ----
    if rank == 0 and world == 1 and not args.no_sweep:
        sweep = []
        try:
            for div in (8, 4, 2, 1):
                Rd = R // div
                sub = pipeline.ReadBatch(reads4[:(Rd * M + 7) // 8 + 4], Rd, M)

                def sub_steps(k):
                    pre = pipeline.seed_pass_begin(fmi, sub, params, 0, None) if (can_pipe and k) else None
                    for i in range(k):
                        nxt = pipeline.seed_pass_begin(fmi, sub, params, (i + 1) & 1, None) if (can_pipe and i + 1 < k) else None
                        pipeline.seed_and_extend(fmi, genome, n, sub, params, None, pre=pre)
                        pre = nxt
                sub_steps(2)
                torch.cuda.synchronize(); s0 = time.perf_counter()
                Ks = 10
                sub_steps(Ks)
                torch.cuda.synchronize(); sdt = (time.perf_counter() - s0) / Ks
                sweep.append({"reads": Rd, "ms_per_step": sdt * 1e3, "reads_per_s": Rd / sdt})
            full = sweep[-1]["ms_per_step"]
            for e in sweep:
                e["x_of_full_batch_ms"] = e["ms_per_step"] / full
                e["ideal"] = e["reads"] / float(R)
            result["strong_scaling_sweep_1gpu"] = sweep
        except Exception as e:
            result["strong_scaling_sweep_1gpu"] = {"error": repr(e)}

    # ---- the same K steps with batch i+1's seed pass on a side stream BESIDE batch i's extension, the seed pass held to a grid of a few hundred workgroups
    #      (2 waves per SIMD: the extension's VALU-bound kernels find room next to it).  The step is faster, but neither kernel then owns the GPU: the
    #      seed launch takes twice as long as alone, which is why the headline and its roofline stay with the one-stream schedule ----
    if rank == 0 and world == 1 and can_pipe and not args.no_overlap_leg and not args.overlap_seed_pass:
        try:
            ov_side = torch.cuda.Stream(device=device)
            old_blocks = params.seed_grid_blocks
            params.seed_grid_blocks = args.overlap_grid_blocks

            def ov_steps(k, tm=None):
                out = None; done = [None, None]
                ov_side.wait_stream(torch.cuda.current_stream(device))
                pre = pipeline.seed_pass_begin(fmi, batch, params, 0, tm, stream=ov_side) if k else None
                for i in range(k):
                    nxt = pipeline.seed_pass_begin(fmi, batch, params, (i + 1) & 1, tm, stream=ov_side, after=done[(i + 1) & 1]) if i + 1 < k else None
                    out = pipeline.seed_and_extend(fmi, genome, n, batch, params, tm, pre=pre)
                    done[i & 1] = torch.cuda.Event(); done[i & 1].record()
                    pre = nxt
                return out
            ov_steps(max(2, args.warmup))
            otm = {}
            torch.cuda.synchronize(); o0 = time.perf_counter()
            Ko = max(args.steps, 12)                                  # (the first batch's seed pass has nothing to run beside: a longer run shows the steady state)
            obs, obp, obrc, onc = ov_steps(Ko, otm)
            torch.cuda.synchronize(); odt = (time.perf_counter() - o0) / Ko
            params.seed_grid_blocks = old_blocks
            result["overlapped_step"] = {"ms_per_step": odt * 1e3, "reads_per_s": R / odt, "steps": Ko, "seed_pass_grid_blocks": args.overlap_grid_blocks,
                                         "kernel_ms_while_sharing_the_gpu": {k: float(np.mean(event_ms(v))) for k, v in otm.items() if k in ("match_both", "extend")},
                                         "results_equal": bool(torch.equal(obs, bs) and torch.equal(obp, bp) and torch.equal(obrc, brc)),
                                         "note": "seed pass of batch i+1 on a side stream beside the extension of batch i (pipeline.seed_pass_begin( stream =, after = )); "
                                                 "not the headline: every kernel's duration is then that of a shared GPU"}
        except Exception as e:
            result["overlapped_step"] = {"error": repr(e)}

    # ---- BASELINE configs 2, 4, 5 at kernel / composition level on the same index (untimed extras; rank 0, one GPU) ----
    if rank == 0 and world == 1 and not args.no_configs:
        try:
            result["configs"] = run_configs(torch, amd, pipeline, fmi, genome, n, device, M, scale=min(1.0, R / 1e7 * 4.0 if R < 2_500_000 else 1.0))
        except Exception as e:                                     # an extra must not take the line down
            result["configs"] = {"error": repr(e)}

    # ---- nvBowtie's own choices (which loci get extended, in which order, when a read gives up) as the C++ host loop over the C ABI, on the
    #      same 10 M reads: seed-hit deques capped at max_hits, select -> locate -> BestScoreStream -> band-31 DP -> score_reduce with the effort
    #      rules, the several-hits-per-read phase once fewer than half the batch is active, reseeding (aligner_best_approx.h:39-207,363-667) ----
    if rank == 0 and world == 1 and not args.no_nvbowtie_mode:
        try:
            stored4 = pack4(reads_sym.flip(1).reshape(-1))                       # nvBowtie stores reads reversed (io::REVERSE)
            sb = pipeline.ReadBatch(stored4, R, M)
            pipeline.nvbowtie_best_approx_host(fmi, genome, n, sb, params)       # warm, at full size: the library's scratch blocks are cached per stream by size
            torch.cuda.synchronize(); t0 = time.perf_counter()
            nb = pipeline.nvbowtie_best_approx_host(fmi, genome, n, sb, params)
            torch.cuda.synchronize(); ndt = time.perf_counter() - t0
            nb_al = nb["best_loc"] >= 0
            both_al = nb_al & aligned
            result["nvbowtie_mode"] = {
                "host": "C++ loop over the C ABI (nvbio-gpl_amd/host/nvbio_amd/best_approx.hpp -> lib/libnvbio_amd_host.so); two counters read per extension pass",
                "reads": R, "ms_per_step": ndt * 1e3, "reads_per_s": R / ndt, "n_extensions": nb["n_extensions"], "extensions_per_read": nb["n_extensions"] / float(R),
                "extension_passes": nb["passes"], "passes_with_several_hits_per_read": nb["multi_passes"], "seeding_passes": nb["seeding_passes"],
                "aligned_fraction": float(nb_al.float().mean()),
                "aligned_by_both": float(both_al.float().mean()),
                "best_score_equals_default_pipeline": float(((nb["best_score"] == bs) & both_al).float().sum() / both_al.float().sum().clamp(min=1)),
                "best_score_below_default_pipeline": float(((nb["best_score"] < bs) & both_al).float().sum() / both_al.float().sum().clamp(min=1)),
                "best_score_above_default_pipeline": float(((nb["best_score"] > bs) & both_al).float().sum() / both_al.float().sum().clamp(min=1)),
                "note": "the default pipeline extends every distinct diagonal once; this mode extends hit by hit as nvBowtie does and stops on its effort rules, so a read can end below the default pipeline's score, never above"}
            del stored4, sb, nb
        except Exception as e:
            result["nvbowtie_mode"] = {"error": repr(e)}

    # ---- the robust-input step (rank 0, one GPU): needs the headline's handle gone ----
    if rank == 0 and world == 1 and not args.no_robust and use_both:
        cpp_inputs = None
        try:
            if not args.no_cpp_host and not args.repeat_family:
                # the batch and the reference as files for the C++ host program (it builds its own index: the handle must go first)
                import tempfile
                tmpd = tempfile.mkdtemp(prefix="nvbio_bench_")
                amd.u32(genome).tofile(os.path.join(tmpd, "genome.u32"))
                amd.u32(reads4).tofile(os.path.join(tmpd, "reads.u32"))
                cpp_inputs = (tmpd, bs.cpu().numpy(), bp.cpu().numpy(), brc.cpu().numpy())
            del batch, reads4, reads_sym, truth_pos, truth_rc, extras
            fmi.close()
            del fmi
            torch.cuda.empty_cache()
            amd.release_scratch()                                   # the library keeps its scratch blocks per stream: the second index needs the room
        except Exception as e:
            result["robust"] = {"error": repr(e)}
        if cpp_inputs is not None:
            # the same step, same reads, as a plain C++ program over the C ABI (no Python, no torch): nvbio-gpl_amd/host/fmmap_amd.cpp
            try:
                import shutil
                import subprocess
                tmpd, hbs, hbp, hbrc = cpp_inputs
                exe = os.path.join(ROOT, "nvbio-gpl_amd", "lib", "fmmap_amd")
                outp = os.path.join(tmpd, "best.bin")
                pr = subprocess.run([exe, "--genome", os.path.join(tmpd, "genome.u32"), "--genome-len", str(n), "--reads", os.path.join(tmpd, "reads.u32"),
                                     "--n-reads", str(R), "--read-len", str(M), "--steps", "5", "--kmer", str(args.kmer), "--out", outp]
                                    + (["--no-canonical"] if not use_both else []), capture_output=True, text=True, timeout=600)
                cj = json.loads([l for l in pr.stdout.splitlines() if l.startswith("{")][-1])
                raw = np.fromfile(outp, dtype=np.uint8)
                cs_ = raw[:4 * R].view(np.int32); cp_ = raw[4 * R:12 * R].view(np.int64); cr_ = raw[12 * R:13 * R]
                cj["results_equal_python_step"] = bool(np.array_equal(cs_, hbs) and np.array_equal(cp_, hbp) and np.array_equal(cr_, hbrc))
                if not cj["results_equal_python_step"]:
                    diff = np.nonzero((cs_ != hbs) | (cp_ != hbp) | (cr_ != hbrc))[0]
                    cj["reads_that_differ"] = int(len(diff))
                    cj["first_differences"] = [[int(r), int(cs_[r]), int(hbs[r]), int(cp_[r]), int(hbp[r]), int(cr_[r]), int(hbrc[r])] for r in diff[:8]]
                cj["note"] = ("the timed step's composition without second best / MAPQ, host side in C++ (hipDeviceSynchronize at both ends of every step, "
                              "no step pipelining); same reads, same reference, its own index build")
                result["cpp_host"] = cj
                shutil.rmtree(tmpd, ignore_errors=True)
            except Exception as e:
                result["cpp_host"] = {"error": repr(e)}
        if "robust" not in result:
            try:
                result["robust"] = run_robust(torch, np, amd, pipeline, args, genome, n, R, M, device, rank, step_ms)
            except Exception as e:
                result["robust"] = {"error": repr(e)}

    if rank == 0:
        print(json.dumps(result), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
