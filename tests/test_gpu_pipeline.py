"""GPU parity of the whole seed-and-extend composition (match -> scan -> expand+locate ->
diagonals -> band-31 Gotoh -> best per read) against the oracle-built CPU path, plus the
size-independent properties used at benchmark scale."""
import importlib

import numpy as np
import pytest

import oracle
from oracle import cpu_pipeline
from util import mutate_reads

pytestmark = pytest.mark.gpu


def test_pipeline_seed_hit_cap_on_repeats(amd, orc):
    """a genome with a 40-copy repeat: without a cap every read from it extends 9 seeds x 40 rows; with
    max_seed_hits = 4 only the first 4 rows of each seed's SA range -- same composition on the oracle"""
    import torch
    pipeline = importlib.import_module("nvbio_gpl_amd.pipeline")
    rng = np.random.default_rng(3)
    G = 600_000
    text = rng.integers(0, 4, G, dtype=np.uint8)
    unit = rng.integers(0, 4, 400, dtype=np.uint8)
    for c in range(40):
        text[20000 + 9000 * c:20400 + 9000 * c] = unit
    hidx = orc.build_index(text)
    genome2 = orc.pack2(text)
    fmi = amd.FMIndex.build(genome2, G, kmer_len=9, sa_int=1)
    R, M = 4000, 150
    starts = rng.integers(0, G - M - 8, R)
    starts[:600] = 20000 + 9000 * rng.integers(0, 40, 600) + rng.integers(0, 240, 600)      # reads inside the repeat
    reads = mutate_reads(rng, text, starts, M)
    rcm = rng.random(R) < 0.5
    reads[rcm] = 3 - reads[rcm][:, ::-1]
    rb = pipeline.ReadBatch(torch.from_numpy(orc.pack4(reads.reshape(-1)).view(np.int32)).cuda(), R, M)
    g_dev = torch.from_numpy(genome2.view(np.int32)).cuda()
    n_c = []
    for cap in (None, 4):
        params = pipeline.SeedExtendParams.end_to_end(max_seed_hits=cap)
        want = cpu_pipeline.seed_and_extend_cpu(orc, hidx, text, G, reads, aln_type=oracle.SEMI_GLOBAL,
                                                scheme=oracle.Scheme(0, 6, 6, -8, -3, -8, -3), max_seed_hits=cap)
        bs, bp, brc, nc = pipeline.seed_and_extend(fmi, g_dev, G, rb, params)
        assert np.array_equal(bs.cpu().numpy(), want[0]) and np.array_equal(bp.cpu().numpy(), want[1])
        assert np.array_equal(brc.cpu().numpy(), want[2])
        assert want[3] <= nc <= want[3] * 1.10
        n_c.append(nc)
    assert n_c[1] < n_c[0] / 2                                  # the cap removed most of the repeat's candidates
    fmi.close()


@pytest.mark.parametrize("mode", ["local", "e2e", "e2e-quals", "fmmap-ed"])
def test_pipeline_equals_cpu_path(amd, orc, mode):
    import torch
    pipeline = importlib.import_module("nvbio_gpl_amd.pipeline")
    rng = np.random.default_rng(21)
    G = 2_000_000
    text = rng.integers(0, 4, G, dtype=np.uint8)
    text[100000:101000] = text[500000:501000]            # a repeat: seeds with two loci
    hidx = orc.build_index(text)
    genome2 = orc.pack2(text)
    fmi = amd.FMIndex.build(genome2, G, kmer_len=10)
    R, M = 20000, 150
    starts = rng.integers(0, G - M - 8, R)
    starts[:50] = rng.integers(100000, 100800, 50)
    starts[50:60] = rng.integers(0, 5, 10)               # clipped windows at the genome start
    starts[60:70] = G - M - 8 - rng.integers(0, 3, 10)   # ... and at the end
    reads = mutate_reads(rng, text, starts, M)
    rcm = rng.random(R) < 0.5
    reads[rcm] = 3 - reads[rcm][:, ::-1]
    reads[rng.random(reads.shape) < 0.001] = 4
    reads[-20:] = rng.integers(0, 4, (20, M))            # unalignable reads
    if mode == "local":
        params = pipeline.SeedExtendParams()
        want = cpu_pipeline.seed_and_extend_cpu(orc, hidx, text, G, reads, traceback_stride=24, traceback_min_score=params.min_score_for(M))
    elif mode == "e2e-quals":                             # end-to-end with the quality ramp (mismatch -2..-6 by base quality)
        quals = rng.integers(0, 50, R * M, dtype=np.uint8)
        params = pipeline.SeedExtendParams.end_to_end(constant_quality=False)
        want = cpu_pipeline.seed_and_extend_cpu(orc, hidx, text, G, reads, aln_type=oracle.SEMI_GLOBAL,
                                                scheme=oracle.Scheme(0, 2, 6, -8, -3, -8, -3), quals=quals, traceback_stride=24, traceback_min_score=params.min_score_for(M))
    elif mode == "fmmap-ed":                              # examples/fmmap: semi-global edit distance (fmmap.cu:346-359)
        params = pipeline.SeedExtendParams(aln_type=oracle.SEMI_GLOBAL, scheme=amd.EditDistanceScheme(), min_score=-15)
        want = cpu_pipeline.seed_and_extend_cpu(orc, hidx, text, G, reads, aln_type=oracle.SEMI_GLOBAL,
                                                scheme=oracle.Scheme(*oracle.ED_SCHEME), traceback_stride=24, traceback_min_score=params.min_score_for(M))
    else:                                                 # nvBowtie default mode, constant quality
        params = pipeline.SeedExtendParams.end_to_end()
        want = cpu_pipeline.seed_and_extend_cpu(orc, hidx, text, G, reads, aln_type=oracle.SEMI_GLOBAL,
                                                scheme=oracle.Scheme(0, 6, 6, -8, -3, -8, -3), traceback_stride=24, traceback_min_score=params.min_score_for(M))
        assert params.min_score_for(150) == -90
    rb = pipeline.ReadBatch(torch.from_numpy(orc.pack4(reads.reshape(-1)).view(np.int32)).cuda(), R, M,
                            quals=torch.from_numpy(quals).cuda() if mode == "e2e-quals" else None)
    g_dev = torch.from_numpy(genome2.view(np.int32)).cuda()
    bs, bp, brc, nc, bwb, bg = pipeline.seed_and_extend(fmi, g_dev, G, rb, params, return_windows=True)
    assert want[3] <= nc <= want[3] * 1.10           # adjacent-duplicate removal may keep a few repeats
    assert np.array_equal(bs.cpu().numpy(), want[0])
    assert np.array_equal(bp.cpu().numpy(), want[1])
    assert np.array_equal(brc.cpu().numpy(), want[2])
    # properties that also hold at full size: reads come back to their locus on the right strand
    ok = (bs.cpu().numpy()[:-20] >= params.min_score_for(150)).mean()
    assert ok > 0.99
    near = np.abs(bp.cpu().numpy()[70:-20] - (starts[70:-20] + M)) <= 40
    assert near.mean() > 0.98
    assert (brc.cpu().numpy()[70:-20][near] == rcm[70:-20][near]).all()
    # traceback of every aligned read's best candidate: CIGARs equal the CPU path's
    ids, tsc, tpos, tsrc, tsnk, tcig, tln = pipeline.traceback_best(g_dev, G, rb, params, bs, brc, bwb, cigar_stride=24, best_pos=bp)
    tb = want[4]
    assert np.array_equal(ids.cpu().numpy(), tb["ids"])
    assert np.array_equal(tsc.cpu().numpy(), tb["scores"]) and np.array_equal(tsc.cpu().numpy(), bs.cpu().numpy()[tb["ids"]])
    assert np.array_equal(tpos.cpu().numpy(), tb["pos"])
    assert np.array_equal(amd.u32(tsrc), tb["sources"]) and np.array_equal(amd.u32(tsnk), tb["sinks"])
    assert np.array_equal(amd.u32(tln), tb["lens"]) and np.array_equal(tcig.cpu().numpy().view(np.uint16), tb["cigars"])
    assert int(tb["lens"].max()) <= 24
    # the traced alignment ends where the scoring pass said it would
    assert np.array_equal(bwb.cpu().numpy()[tb["ids"]] + tb["sinks"][:, 0].astype(np.int64), bp.cpu().numpy()[tb["ids"]])
    # reads with one indel carry exactly one gap element; most reads are a single run of matches
    one_run = (tb["lens"] == 1).mean()
    assert one_run > 0.4
    # the all-reads form (batch built on the device from the per-read best keys, no compaction): the same alignments, row r = read r
    ex = {}
    o = pipeline.seed_and_extend(fmi, g_dev, G, rb, params, return_windows=True, extras=ex)
    assert torch.equal(o[4], bwb) and torch.equal(o[5], bg)
    asc, apos, asrc, asnk, acig, aln = pipeline.traceback_best_all(g_dev, G, rb, params, ex["best_keys"], bwb, cigar_stride=24)
    sel = tb["ids"]
    traced = np.zeros(R, dtype=bool); traced[sel] = True
    assert np.array_equal(amd.u32(aln) > 0, traced)
    assert np.array_equal(asc.cpu().numpy()[sel], tb["scores"]) and np.array_equal(apos.cpu().numpy()[sel], tb["pos"])
    assert np.array_equal(amd.u32(asrc)[sel], tb["sources"]) and np.array_equal(amd.u32(asnk)[sel], tb["sinks"])
    assert np.array_equal(amd.u32(aln)[sel], tb["lens"]) and np.array_equal(acig.cpu().numpy().view(np.uint16)[sel], tb["cigars"])
    assert (amd.u32(asnk)[~traced] == 0xFFFFFFFF).all()
    fmi.close()


def test_paired_end_equals_cpu_path(amd, orc):
    """FR pairs with insert ~ N(350, 50): each mate anchored in turn, the other scored by full-matrix DP inside the
    opposite-mate window (nvBowtie's rule), best pair per read -- equal to the same composition on the oracle"""
    import torch
    pipeline = importlib.import_module("nvbio_gpl_amd.pipeline")
    rng = np.random.default_rng(9)
    G = 1_000_000
    text = rng.integers(0, 4, G, dtype=np.uint8)
    hidx = orc.build_index(text)
    genome2 = orc.pack2(text)
    fmi = amd.FMIndex.build(genome2, G, kmer_len=10, sa_int=1)
    R, M = 3000, 150
    ins = np.clip(rng.normal(350, 50, R), 160, 500).astype(np.int64)
    left = rng.integers(0, G - 520, R)
    left[:20] = rng.integers(0, 30, 20); left[20:40] = G - 520 - rng.integers(0, 10, 20)      # fragments at both genome ends
    m1 = mutate_reads(rng, text, left, M)                                                     # forward mate at the fragment start
    m2 = mutate_reads(rng, text, left + ins - M, M)
    m2 = (3 - m2[:, ::-1]).astype(np.uint8)                                                   # reverse mate at its end (FR)
    swap = rng.random(R) < 0.5                                                                # the fragment comes from either strand
    m1[swap], m2[swap] = m2[swap].copy(), m1[swap].copy()
    m2[-30:] = rng.integers(0, 4, (30, M))                                                    # mates that belong nowhere
    params = pipeline.SeedExtendParams.end_to_end()
    want = cpu_pipeline.paired_end_cpu(orc, hidx, text, G, m1, m2, oracle.Scheme(0, 6, 6, -8, -3, -8, -3), params.min_score_for,
                                       oracle.SEMI_GLOBAL, cigar_stride=24)
    g_dev = torch.from_numpy(genome2.view(np.int32)).cuda()
    b1 = pipeline.ReadBatch(torch.from_numpy(orc.pack4(m1.reshape(-1)).view(np.int32)).cuda(), R, M)
    b2 = pipeline.ReadBatch(torch.from_numpy(orc.pack4(m2.reshape(-1)).view(np.int32)).cuda(), R, M)
    got = pipeline.paired_end(fmi, g_dev, G, b1, b2, params, cigar_stride=24)
    for k in ("anchor", "pair_score", "score1", "score2", "pos1", "pos2", "rc1", "rc2", "begin1", "begin2"):
        assert np.array_equal(got[k].cpu().numpy().astype(np.int64), want[k].astype(np.int64)), k
    for m in (1, 2):        # CIGARs of both mates of every chosen pair (anchor: banded traceback, opposite: full matrix)
        assert np.array_equal(amd.u32(got["cigar_lens%d" % m]), want["cigar_lens%d" % m]), m
        assert np.array_equal(got["cigars%d" % m].cpu().numpy().view(np.uint16), want["cigars%d" % m]), m
    assert int(want["cigar_lens1"].max()) <= 24 and (want["cigar_lens2"][want["anchor"] >= 0] >= 1).all()
    paired = want["anchor"] >= 0
    assert paired[:-30].mean() > 0.97 and not paired[-30:].any()
    # concordant FR pairs: opposite strands, ends within the fragment
    ok = paired & (want["rc1"] != want["rc2"]) & (np.abs(want["pos1"] - want["pos2"]) <= 500)
    assert ok.sum() >= 0.99 * paired.sum()
    fmi.close()


def test_mapq_kernel_covers_every_branch(amd, orc):
    """nvbio_mapq against the oracle's restatement of BowtieMapq2 / BowtieMapq3 over a grid of (best, second) scores that
    reaches every branch of mapq.h, for the end-to-end (monotone) and the local scoring ranges"""
    import torch
    for monotone, perfect, minimum in ((True, 0, -90), (False, 300, 50), (True, 0, -61), (False, 200, 53)):
        best = np.arange(minimum - 3, perfect + 1, dtype=np.int64)
        sec = np.arange(minimum - 3, perfect + 1, 7, dtype=np.int64)
        B, S = np.meshgrid(best, sec, indexing="ij")
        keep = S <= B
        B, S = B[keep], S[keep]
        has = (np.arange(len(B)) % 5) != 0                      # every fifth pair: no second alignment
        bk = ((B + (1 << 20)) << 34) | 12345
        sk = np.where(has, ((S + (1 << 20)) << 34) | (1 << 33) | 999, 0)
        for version in (2, 3):
            q, ss = amd.mapq(torch.from_numpy(bk).cuda(), torch.from_numpy(sk).cuda(), perfect, minimum, monotone, version)
            want = np.array([orc.mapq(version, monotone, perfect, minimum, int(b), bool(h), int(s)) for b, s, h in zip(B, S, has)], dtype=np.uint8)
            assert np.array_equal(q.cpu().numpy(), want), (monotone, version)
            assert np.array_equal(ss.cpu().numpy()[has], S[has].astype(np.int32))
            assert len(np.unique(want)) >= (12 if version == 2 else 8)


@pytest.mark.parametrize("mode", ["e2e", "local"])
def test_pipeline_second_best_and_mapq(amd, orc, mode):
    """nvBowtie's best / second-best bookkeeping (score_reduce over the candidates in descending key order) and the mapping
    quality on a genome with diverged repeats: exact copies (second = best), copies with a few substitutions (second < best),
    tandem copies closer than read_len / 2 (not distinct: no second), copies on the other strand"""
    import torch
    pipeline = importlib.import_module("nvbio_gpl_amd.pipeline")
    rng = np.random.default_rng(77)
    G = 1_000_000
    text = rng.integers(0, 4, G, dtype=np.uint8)
    unit = text[50000:50600].copy()
    text[200000:200600] = unit                                        # exact copy
    div = unit.copy(); div[rng.choice(600, 12, replace=False)] = rng.integers(0, 4, 12)
    text[300000:300600] = div                                         # diverged copy
    text[400000:400600] = 3 - unit[::-1]                              # reverse-complement copy
    text[50640:50940] = unit[:300]                                    # tandem copy 40 bp after the original ends
    text[600000:600060] = np.tile(text[600000:600003], 20)
    hidx = orc.build_index(text)
    genome2 = orc.pack2(text)
    fmi = amd.FMIndex.build(genome2, G, kmer_len=10, sa_int=1)
    R, M = 6000, 150
    starts = rng.integers(0, G - M - 8, R)
    starts[:1500] = 50000 + rng.integers(0, 450, 1500)
    starts[1500:2000] = 300000 + rng.integers(0, 450, 500)
    starts[2000:2300] = 400000 + rng.integers(0, 450, 300)
    reads = mutate_reads(rng, text, starts, M)
    rcm = rng.random(R) < 0.5
    reads[rcm] = 3 - reads[rcm][:, ::-1]
    reads[-20:] = rng.integers(0, 4, (20, M))
    if mode == "e2e":
        params = pipeline.SeedExtendParams.end_to_end()
        osch, otype, perfect, mono = oracle.Scheme(0, 6, 6, -8, -3, -8, -3), oracle.SEMI_GLOBAL, 0, True
    else:
        params = pipeline.SeedExtendParams()
        osch, otype, perfect, mono = oracle.Scheme(2, 2, 6, -8, -3, -8, -3), oracle.LOCAL, 2 * M, False
    params.mapq = True
    ms = params.min_score_for(M)
    want = cpu_pipeline.seed_and_extend_cpu(orc, hidx, text, G, reads, aln_type=otype, scheme=osch,
                                            second=dict(min_score=ms, perfect_score=perfect, monotone=mono, version=2))
    rb = pipeline.ReadBatch(torch.from_numpy(orc.pack4(reads.reshape(-1)).view(np.int32)).cuda(), R, M)
    g_dev = torch.from_numpy(genome2.view(np.int32)).cuda()
    extras = {}
    bs, bp, brc, nc = pipeline.seed_and_extend(fmi, g_dev, G, rb, params, extras=extras)
    assert np.array_equal(bs.cpu().numpy(), want[0]) and np.array_equal(bp.cpu().numpy(), want[1])
    x = want[4]
    assert np.array_equal(extras["second_score"].cpu().numpy(), x["second_score"])
    sk = extras["second"].cpu().numpy()
    has = sk != 0
    assert np.array_equal(has, x["second_pos"] >= 0)
    assert np.array_equal((sk & ((1 << 33) - 1))[has], x["second_pos"][has]) and np.array_equal(((sk >> 33) & 1)[has], x["second_rc"][has])
    assert np.array_equal(extras["mapq"].cpu().numpy(), x["mapq"])
    q = x["mapq"]
    # the repeats did their job: reads with a second alignment and a range of qualities; unique reads keep high ones
    assert has[:2300].mean() > 0.8 and has[2300:-20].mean() < 0.05
    assert len(np.unique(q)) >= 6 and (q[2300:-20] >= 20).mean() > 0.95 and (q[:1500] <= 1).mean() > 0.5
    fmi.close()


def test_pipeline_ragged_reads_with_qualities_and_a_repeat_family(amd, orc):
    """what the benchmark's friendly batch is not: reads of different lengths (each seeded at its own interval, scored against its own
    threshold), per-base qualities under nvBowtie's default ramp (mismatch 2..6: the quality-aware first pass), and a repeat family
    whose k-mers the canonical table cannot answer (deferred searches, seed-hit cap).  Every read is independent, so the ragged batch
    must give each read what the oracle's uniform composition gives a batch of that read's length."""
    import torch
    pipeline = importlib.import_module("nvbio_gpl_amd.pipeline")
    rng = np.random.default_rng(2026)
    G = 1_500_000
    text = rng.integers(0, 4, G, dtype=np.uint8)
    unit = rng.integers(0, 4, 300, dtype=np.uint8)
    for c in range(60):
        text[20000 + 9000 * c:20300 + 9000 * c] = unit
    hidx = orc.build_index(text)
    genome2 = orc.pack2(text)
    fmi = amd.FMIndex.build(genome2, G, kmer_len=15, sa_int=1, table_flags=amd.FM_TABLE_CANONICAL_WIDE)
    assert fmi.canonical
    R, Mmax = 6000, 150
    lens = rng.choice(np.array([100, 101, 117, 125, 136, 149, 150]), R)
    starts = rng.integers(0, G - Mmax - 8, R)
    starts[:500] = 20000 + 9000 * rng.integers(0, 60, 500) + rng.integers(0, 150, 500)       # reads inside the family
    full = mutate_reads(rng, text, starts, Mmax)
    rcm = rng.random(R) < 0.5
    reads = []
    for r in range(R):
        x = full[r, :lens[r]]
        x = ((3 - x[::-1]) if rcm[r] else x).copy()
        x[rng.random(len(x)) < 0.0005] = 4
        reads.append(x)
    offs = np.zeros(R + 1, dtype=np.int64); offs[1:] = np.cumsum(lens)
    flat = np.concatenate(reads).astype(np.uint8)
    quals = rng.choice(np.array([2, 12, 23, 37, 40], dtype=np.uint8), len(flat), p=[0.05, 0.1, 0.15, 0.5, 0.2])
    params = pipeline.SeedExtendParams.end_to_end(constant_quality=False, max_seed_hits=4)
    params.mapq = True
    rb = pipeline.ReadBatch(torch.from_numpy(orc.pack4(flat).view(np.int32)).cuda(), R, Mmax, quals=torch.from_numpy(quals).cuda(),
                            offsets=torch.from_numpy(offs.astype(np.int32)).cuda())
    g_dev = torch.from_numpy(genome2.view(np.int32)).cuda()
    scheme = oracle.Scheme(0, 2, 6, -8, -3, -8, -3)
    results = {}
    for defer, one_call in ((True, True), (False, True), (True, False)):
        for algo in (0, amd.ALN_NO_UNGAPPED_SCORE):
            params.defer_heavy, params.algo_flags, params.one_call_residuals = defer, algo, one_call
            ex = {}
            bs, bp, brc, nc = pipeline.seed_and_extend(fmi, g_dev, G, rb, params, extras=ex)
            results[(defer, one_call, algo)] = (bs.cpu().numpy(), bp.cpu().numpy(), brc.cpu().numpy(), ex["mapq"].cpu().numpy(), ex["second_score"].cpu().numpy())
    base = results[(True, True, 0)]
    for key, val in results.items():
        for a, b in zip(base, val):
            assert np.array_equal(a, b), key
    itab = params.interval_table(Mmax)
    checked = 0
    for ln in np.unique(lens):
        grp = np.nonzero(lens == ln)[0]
        assert int(1 + 1.15 * np.sqrt(ln)) == int(itab[ln])
        sub = np.stack([reads[r] for r in grp]).astype(np.uint8)
        q = np.concatenate([quals[offs[r]:offs[r + 1]] for r in grp])
        ms = params.min_score_for(int(ln))
        want = cpu_pipeline.seed_and_extend_cpu(orc, hidx, text, G, sub, aln_type=oracle.SEMI_GLOBAL, scheme=scheme, quals=q, max_seed_hits=4,
                                                second=dict(min_score=ms, perfect_score=0, monotone=True, version=2))
        assert np.array_equal(base[0][grp], want[0]), ln
        assert np.array_equal(base[1][grp], want[1]), ln
        assert np.array_equal(base[2][grp], want[2]), ln
        assert np.array_equal(base[3][grp], want[4]["mapq"]), ln
        assert np.array_equal(base[4][grp], want[4]["second_score"]), ln
        checked += len(grp)
    assert checked == R
    aligned = np.array([base[0][r] >= params.min_score_for(int(lens[r])) for r in range(R)])
    assert aligned[500:].mean() > 0.97
    fmi.close()
