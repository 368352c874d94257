"""GPU parity: batched banded Gotoh through the C-ABI vs the golden vectors of the reference and
vs the oracle.  Integer scores and sinks must be exactly equal."""
import numpy as np
import pytest

import oracle
from util import mutate_reads

pytestmark = pytest.mark.gpu


def _scheme(amd, v):
    return amd.GotohScheme(*[int(x) for x in v])


@pytest.mark.parametrize("band", [3, 7, 15, 31])
def test_banded_golden_bytes(amd, dp_golden, band):
    """every golden pair, one symbol per byte (N = 4 in patterns), with and without qualities"""
    g = dp_golden
    S = len(g["schemes"])
    n = len(g["pat_off"]) - 1
    bi = list(g["bands"]).index(band)
    for si in range(S):
        for hq in (0, 1):
            cases = np.array([i for i in range(n) if i % S == si and int(g["has_quals"][i]) == hq], dtype=np.uint32)
            if len(cases) == 0:
                continue
            batch = amd.AlignmentBatch(g["pats"], 8, g["pat_off"], g["txts"], 8, g["txt_off"][cases],
                                       g["txt_off"][cases + 1], quals=g["quals"] if hq else None, read_id=cases)
            for typ in range(3):
                sc, sk = amd.batch_banded_alignment_score(band, amd.make_gotoh_aligner(typ, _scheme(amd, g["schemes"][si])), batch)
                want = g["banded"][cases, bi, typ]
                assert np.array_equal(sc.cpu().numpy().astype(np.int64), want[:, 1]), (band, si, hq, typ)
                assert np.array_equal(amd.u32(sk).astype(np.int64), want[:, 2:4]), (band, si, hq, typ)


def test_banded_random_packed(amd, orc):
    """nvBowtie-shaped batch: 4-bit reads (fw / reversed / complemented), windows on a 2-bit genome
    including windows clipped at both genome ends, quality-dependent mismatches"""
    rng = np.random.default_rng(3)
    G = 300000
    text = rng.integers(0, 4, G, dtype=np.uint8)
    R, M = 6000, 150
    starts = rng.integers(0, G - M - 8, R)
    reads = mutate_reads(rng, text, starts, M)
    reads[rng.random(reads.shape) < 0.002] = 4                  # a few N
    lens = np.full(R, M); lens[::7] = rng.integers(30, 150, len(lens[::7]))
    roffs = np.zeros(R + 1, dtype=np.uint32); roffs[1:] = np.cumsum(lens)
    flat = np.concatenate([reads[k, :lens[k]] for k in range(R)])
    quals = rng.integers(0, 64, len(flat), dtype=np.uint8)
    J = 20000
    rid = rng.integers(0, R, J).astype(np.uint32)
    flags = rng.integers(0, 4, J).astype(np.uint8)
    flags[:J // 2] = 0
    g_pos = starts[rid].astype(np.int64) + rng.integers(-3, 4, J)
    g_pos[::50] = rng.integers(0, 10, len(g_pos[::50]))         # clipped at the genome start
    g_pos[1::50] = G - rng.integers(100, 170, len(g_pos[1::50]))   # clipped at the genome end
    g_pos = np.clip(g_pos, 0, G - 1)
    for band in (31, 15, 7, 3):
        # BestScoreStream window (nvBowtie/bowtie2/cuda/score_inl.h:100-106)
        wb = np.where(g_pos > band // 2, g_pos - band // 2, 0).astype(np.uint32)
        we = np.minimum(wb + band + lens[rid], G).astype(np.uint32)
        we = np.maximum(we, wb + band - 1)                      # keep N >= BAND-1 (undefined in the reference below that)
        we = np.minimum(we, G).astype(np.uint32)
        ok = (we - wb) >= band - 1
        sel = np.nonzero(ok)[0]
        for typ, sv in ((oracle.LOCAL, (2, 2, 6, -8, -3, -8, -3)), (oracle.SEMI_GLOBAL, (0, 2, 6, -8, -3, -8, -3)),
                        (oracle.GLOBAL, (1, 3, 3, -15, -4, -11, -2))):
            for use_q in (True, False):
                batch = amd.AlignmentBatch(orc.pack4(flat), 4, roffs, orc.pack2(text), 2, wb[sel], we[sel],
                                           quals=quals if use_q else None, read_id=rid[sel], flags=flags[sel])
                sc, sk = amd.batch_banded_alignment_score(band, amd.make_gotoh_aligner(typ, _scheme(amd, sv)), batch)
                wsc, wsk = orc.banded_gotoh_packed_batch(band, typ, oracle.Scheme(*sv), orc.pack4(flat), roffs,
                                                         orc.pack2(text), wb[sel], we[sel], read_id=rid[sel],
                                                         flags=flags[sel], quals=quals if use_q else None)
                assert np.array_equal(sc.cpu().numpy(), wsc), (band, typ, use_q)
                assert np.array_equal(amd.u32(sk), wsk), (band, typ, use_q)


def test_quality_ramp_all_values(amd, orc):
    """the float->int mismatch ramp (scoring.h:84-88) for every quality 0..255 and several ramps"""
    pat = np.array([0], dtype=np.uint8)
    txt = np.concatenate([[1], np.zeros(40, dtype=np.uint8)]).astype(np.uint8)
    for mm_min, mm_max in ((2, 6), (0, 30), (1, 7), (3, 3), (6, 2), (0, 255)):
        q = np.arange(256, dtype=np.uint8)
        pats = np.zeros(256, dtype=np.uint8)
        roffs = np.arange(257, dtype=np.uint32)
        batch = amd.AlignmentBatch(pats, 8, roffs, txt, 8, np.zeros(256, dtype=np.uint32),
                                   np.full(256, 31, dtype=np.uint32), quals=q)
        sv = (5, mm_min, mm_max, -100, -100, -100, -100)
        sc, _ = amd.batch_banded_alignment_score(3, amd.make_gotoh_aligner(oracle.GLOBAL, _scheme(amd, sv)), batch)
        want = [orc.banded_gotoh(3, oracle.GLOBAL, oracle.Scheme(*sv), pat, txt[:31], np.array([v], dtype=np.uint8))[1]
                for v in range(256)]
        assert list(sc.cpu().numpy()) == want, (mm_min, mm_max)


def test_banded_edge_cases(amd, orc):
    """text shorter than the pattern (nothing reported), empty batch, unsupported band"""
    txt = np.zeros(64, dtype=np.uint8)
    pats = np.zeros(80, dtype=np.uint8)
    roffs = np.array([0, 40, 80], dtype=np.uint32)
    batch = amd.AlignmentBatch(pats, 8, roffs, txt, 8, np.array([0, 0], dtype=np.uint32),
                               np.array([35, 64], dtype=np.uint32))
    al = amd.make_gotoh_aligner(oracle.LOCAL, amd.SimpleGotohScheme(2, -1, -2, -1))
    sc, sk = amd.batch_banded_alignment_score(31, al, batch)
    assert sc.cpu().numpy()[0] == amd.SCORE_MIN and list(amd.u32(sk)[0]) == [0xFFFFFFFF, 0xFFFFFFFF]
    assert sc.cpu().numpy()[1] == 80
    with pytest.raises(amd.NvbioError):
        amd.batch_banded_alignment_score(9, al, batch)
    empty = amd.AlignmentBatch(pats, 8, roffs, txt, 8, np.zeros(0, dtype=np.uint32), np.zeros(0, dtype=np.uint32))
    sc, sk = amd.batch_banded_alignment_score(31, al, empty)
    assert sc.numel() == 0


@pytest.mark.parametrize("typ", ["LOCAL", "SEMI_GLOBAL", "GLOBAL"])
def test_packed_band31_kernel_equals_int32_and_oracle(amd, orc, typ):
    """the 16-bit two-alignments-per-lane kernel (picked when max_read_len is given and every score
    fits) against the oracle: odd batch sizes, ragged lengths, pairs of different length, short
    texts (N < M on one half of a pair), reversed/complemented reads, qualities, clipped windows"""
    rng = np.random.default_rng(77)
    G = 200000
    text = rng.integers(0, 4, G, dtype=np.uint8)
    R = 5001
    lens = rng.integers(20, 151, R); lens[::3] = 150
    roffs = np.zeros(R + 1, dtype=np.uint32); roffs[1:] = np.cumsum(lens)
    starts = rng.integers(0, G - 200, R)
    flat = np.concatenate([text[s:s + l] for s, l in zip(starts, lens)]).copy()
    flat[rng.random(len(flat)) < 0.03] = rng.integers(0, 5)
    flat[rng.integers(0, len(flat), 300)] = 4
    quals = rng.integers(0, 64, len(flat), dtype=np.uint8)
    J = 20001
    rid = rng.integers(0, R, J).astype(np.uint32)
    flags = rng.integers(0, 4, J).astype(np.uint8)
    flags[: J // 2] = 0
    g_pos = starts[rid].astype(np.int64) + rng.integers(-3, 4, J)
    g_pos[::40] = rng.integers(0, 10, len(g_pos[::40]))
    g_pos[1::40] = G - rng.integers(20, 170, len(g_pos[1::40]))
    g_pos = np.clip(g_pos, 0, G - 1)
    wb = np.where(g_pos > 15, g_pos - 15, 0).astype(np.uint32)
    we = np.minimum(wb + 31 + lens[rid], G).astype(np.uint32)
    we = np.minimum(np.maximum(we, wb + 30), G).astype(np.uint32)
    sel = np.nonzero((we - wb) >= 30)[0]
    assert (we[sel] - wb[sel] < lens[rid[sel]]).any()           # some N < M jobs are in
    typ = getattr(oracle, typ)
    # nvBowtie local() / a constant-mismatch scheme / asymmetric gaps; for the end-to-end types also nvBowtie's
    # e2e defaults (match 0, scoring_inl.h:49-70) with the quality ramp and with the constant -6
    schemes = [(2, 2, 6, -8, -3, -8, -3), (1, 3, 3, -11, -4, -11, -4), (3, 0, 9, -5, -5, -2, -1)]
    if typ != oracle.LOCAL:
        schemes += [(0, 2, 6, -8, -3, -8, -3), (0, 6, 6, -8, -3, -8, -3)]
    for sv in schemes:
        for use_q in (True, False):
            kw = dict(quals=quals if use_q else None, read_id=rid[sel], flags=flags[sel])
            wsc, wsk = orc.banded_gotoh_packed_batch(31, typ, oracle.Scheme(*sv), orc.pack4(flat), roffs,
                                                     orc.pack2(text), wb[sel], we[sel], **kw)
            for hint in (150, 0):                               # packed kernel / int32 kernel
                batch = amd.AlignmentBatch(orc.pack4(flat), 4, roffs, orc.pack2(text), 2, wb[sel], we[sel],
                                           max_read_len=hint, **kw)
                sc, sk = amd.batch_banded_alignment_score(31, amd.make_gotoh_aligner(typ, _scheme(amd, sv)), batch)
                assert np.array_equal(sc.cpu().numpy(), wsc), (sv, use_q, hint)
                assert np.array_equal(amd.u32(sk), wsk), (sv, use_q, hint)
    # a scheme whose scores do not fit 10 bits must fall back to the int32 kernel and still be exact
    sv = (9, 2, 60, -8, -3, -8, -3)
    batch = amd.AlignmentBatch(orc.pack4(flat), 4, roffs, orc.pack2(text), 2, wb[sel], we[sel], max_read_len=150,
                               read_id=rid[sel], flags=flags[sel])
    sc, sk = amd.batch_banded_alignment_score(31, amd.make_gotoh_aligner(typ, _scheme(amd, sv)), batch)
    wsc, wsk = orc.banded_gotoh_packed_batch(31, typ, oracle.Scheme(*sv), orc.pack4(flat), roffs, orc.pack2(text),
                                             wb[sel], we[sel], read_id=rid[sel], flags=flags[sel])
    assert np.array_equal(sc.cpu().numpy(), wsc) and np.array_equal(amd.u32(sk), wsk)


def test_banded_edit_distance_golden(amd, dp_golden, ed_golden):
    """make_edit_distance_aligner through the banded kernels == the reference's EditDistanceAligner (fmmap, nvBowtie
    --scoring ed), every band and type, on the reference's own outputs"""
    g, e = dp_golden, ed_golden
    n = len(g["pat_off"]) - 1
    cases = np.arange(n, dtype=np.uint32)
    batch = amd.AlignmentBatch(g["pats"], 8, g["pat_off"], g["txts"], 8, g["txt_off"][cases], g["txt_off"][cases + 1], read_id=cases)
    checked = 0
    for bi, band in enumerate(e["bands"]):
        for typ in range(3):
            sc, sk = amd.batch_banded_alignment_score(int(band), amd.make_edit_distance_aligner(typ), batch)
            sc, sk = sc.cpu().numpy().astype(np.int64), sk.cpu().numpy().astype(np.int64)
            want = e["ed"][:, bi, typ]
            pin = want[:, 0] >= 0
            assert np.array_equal(sc[pin], want[pin, 1]) and np.array_equal(sk[pin], want[pin, 2:4]), (band, typ)
            checked += int(pin.sum())
    assert checked > 4000


def test_banded_smith_waterman_golden(amd, orc, dp_golden, sw_golden):
    """SmithWatermanAligner through nvbio_banded_sw_score == the reference's, every band and type, on the reference's own
    outputs; then unequal deletion / insertion costs (the int32 kernel with separate E and F terms) and the packed 16-bit
    route (deletion == insertion on 4-bit reads / 2-bit text) against the oracle's own restatement of sw_banded_inl.h"""
    g, w = dp_golden, sw_golden
    n = len(g["pat_off"]) - 1
    cases = np.arange(n, dtype=np.uint32)
    batch = amd.AlignmentBatch(g["pats"], 8, g["pat_off"], g["txts"], 8, g["txt_off"][cases], g["txt_off"][cases + 1], read_id=cases)
    checked = 0
    for si, sw in enumerate(w["schemes"]):
        for bi, band in enumerate(w["bands"]):
            for typ in range(3):
                al = amd.make_smith_waterman_aligner(typ, amd.SimpleSmithWatermanScheme(*[int(v) for v in sw]))
                sc, sk = amd.batch_banded_alignment_score(int(band), al, batch)
                sc, sk = sc.cpu().numpy().astype(np.int64), sk.cpu().numpy().astype(np.int64)
                want = w["bsw"][:, si, bi, typ]
                pin = want[:, 0] >= 0
                assert np.array_equal(sc[pin], want[pin, 1]) and np.array_equal(sk[pin], want[pin, 2:4]), (si, band, typ)
                checked += int(pin.sum())
    assert checked > 8000
    rng = np.random.default_rng(5)
    G = 200000
    text = rng.integers(0, 4, G, dtype=np.uint8)
    R, M = 4000, 120
    starts = rng.integers(20, G - M - 40, R)
    reads = mutate_reads(rng, text, starts, M)
    roffs = (np.arange(R + 1) * M).astype(np.uint32)
    for band in (31, 15):
        wb = (starts - band // 2).astype(np.uint32); we = (wb + band + M).astype(np.uint32)
        batch = amd.AlignmentBatch(orc.pack4(reads.reshape(-1)), 4, roffs, orc.pack2(text), 2, wb, we, max_read_len=M)
        for sw in ((2, -3, -5, -2), (1, -2, -1, -4), (0, -1, -1, -1), (2, -6, -4, -4)):
            for typ in range(3):
                sc, sk = amd.batch_banded_alignment_score(band, amd.make_smith_waterman_aligner(typ, amd.SimpleSmithWatermanScheme(*sw)), batch)
                sc, sk = sc.cpu().numpy(), amd.u32(sk)
                for k in range(0, R, 7):
                    ok, s, snk = orc.banded_sw(band, typ, sw, reads[k], text[wb[k]:we[k]])
                    assert (int(sc[k]), int(sk[k, 0]), int(sk[k, 1])) == (s, snk[0], snk[1]), (band, sw, typ, k)


def test_best2_sink_golden(amd, dp_golden, best2_golden):
    """nvbio_banded_gotoh_score_best2 / nvbio_full_gotoh_score_best2 == the reference's DP reporting into aln::Best2Sink, every
    band, type, blocking and distinct distance, with and without qualities / minimum scores, on the reference's own outputs"""
    g, w = dp_golden, best2_golden
    S = len(g["schemes"])
    n = len(g["pat_off"]) - 1
    lens_p = np.diff(g["pat_off"]); lens_t = np.diff(g["txt_off"])
    checked = 0
    for si in range(S):
        for hq in (0, 1):
            cases = np.array([i for i in range(n) if i % S == si and int(g["has_quals"][i]) == hq], dtype=np.uint32)
            if len(cases) == 0:
                continue
            batch = amd.AlignmentBatch(g["pats"], 8, g["pat_off"], g["txts"], 8, g["txt_off"][cases], g["txt_off"][cases + 1],
                                       quals=g["quals"] if hq else None, read_id=cases)
            for typ in range(3):
                al = amd.make_gotoh_aligner(typ, _scheme(amd, g["schemes"][si]))
                for di, dist in enumerate(w["dists"]):
                    for bi, band in enumerate(w["bands"]):
                        out = amd.batch_banded_alignment_score_best2(int(band), al, batch, int(dist))
                        want = w["banded"][cases, di, bi, typ]
                        pin = want[:, 0] >= 0
                        got = np.stack([out[0].cpu().numpy().astype(np.int64), amd.u32(out[1])[:, 0], amd.u32(out[1])[:, 1],
                                        out[2].cpu().numpy().astype(np.int64), amd.u32(out[3])[:, 0], amd.u32(out[3])[:, 1]], axis=1)
                        assert np.array_equal(got[pin], want[pin, 1:]), (si, hq, typ, dist, band)
                        checked += int(pin.sum())
                    for blk in (0, 1):
                        for v in (0, 1):
                            out = amd.batch_alignment_score_best2(al, batch, int(lens_p.max()), int(lens_t.max()), int(dist), text_blocking=bool(blk),
                                                                  min_scores=g["min_scores"][cases] if v else None)
                            want = w["full"][cases, di, blk, typ, v]
                            got = np.stack([out[0].cpu().numpy().astype(np.int64), amd.u32(out[1])[:, 0], amd.u32(out[1])[:, 1],
                                            out[2].cpu().numpy().astype(np.int64), amd.u32(out[3])[:, 0], amd.u32(out[3])[:, 1]], axis=1)
                            assert np.array_equal(got, want[:, 1:]), (si, hq, typ, dist, blk, v)
                            checked += len(cases)
    assert checked > 15000


def test_end_to_end_shortcut_edge_cases(amd, orc):
    """the ungapped shortcut and its second chance (U* <= G but only single-gap, mismatch-free alignments could reach
    it): reads with 0-3 substitutions, and reads whose only difference is a 1-4 bp indel a few bases from either end --
    there the ungapped diagonal has two or three mismatches while a gapped alignment scores higher, so the DP must be
    taken -- under several schemes (different admissible gap lengths), windows at full width and clipped"""
    rng = np.random.default_rng(91)
    G = 400000
    text = rng.integers(0, 4, G, dtype=np.uint8)
    text[5000:5300] = np.tile(text[5000:5003], 100)                 # period-3 repeat: shifted diagonals match too
    M = 150
    reads, g_pos = [], []
    def locus():
        return int(rng.integers(100, G - 400))
    for k in range(4):                                              # substitutions only
        for _ in range(300):
            p = locus(); r = text[p:p + M].copy()
            pos = rng.choice(M, k, replace=False)
            r[pos] = (r[pos] + 1 + rng.integers(0, 3, k)) % 4
            reads.append(r); g_pos.append(p)
    for _ in range(1500):                                           # one indel near an end, nothing else
        p = locus(); g = int(rng.integers(1, 5)); e = int(rng.integers(1, 7))
        at = e if rng.random() < 0.5 else M - e
        src = text[p:p + M + 8].copy()
        if rng.random() < 0.5:
            r = np.concatenate([src[:at], src[at + g:]])[:M]        # deletion from the read
        else:
            r = np.concatenate([src[:at], rng.integers(0, 4, g, dtype=np.uint8), src[at:]])[:M]
        reads.append(r.astype(np.uint8)); g_pos.append(p)
    for _ in range(300):                                            # inside the repeat
        p = int(rng.integers(5000, 5100)); r = text[p:p + M].copy()
        k = int(rng.integers(0, 3)); pos = rng.choice(M, k, replace=False); r[pos] = (r[pos] + 1) % 4
        reads.append(r); g_pos.append(p)
    R = len(reads)
    flat = np.concatenate(reads)
    roffs = (np.arange(R + 1) * M).astype(np.uint32)
    g_pos = np.array(g_pos, dtype=np.int64) + rng.integers(-3, 4, R)
    wb = (g_pos - 15).astype(np.uint32)
    we = (wb + 31 + M).astype(np.uint32)
    we[::17] -= rng.integers(1, 25, len(we[::17])).astype(np.uint32)    # clipped windows (N < M + 30)
    for sv in ((0, 6, 6, -8, -3, -8, -3), (0, 2, 2, -5, -1, -5, -1), (0, 4, 4, -6, -6, -6, -6), (0, 3, 3, -3, -3, -9, -1)):
        wsc, wsk = orc.banded_gotoh_packed_batch(31, oracle.SEMI_GLOBAL, oracle.Scheme(*sv), orc.pack4(flat), roffs, orc.pack2(text), wb, we)
        batch = amd.AlignmentBatch(orc.pack4(flat), 4, roffs, orc.pack2(text), 2, wb, we, max_read_len=M)
        sc, sk = amd.batch_banded_alignment_score(31, amd.make_gotoh_aligner(oracle.SEMI_GLOBAL, _scheme(amd, sv)), batch)
        bad = np.nonzero(sc.cpu().numpy() != wsc)[0]
        assert len(bad) == 0, (sv, bad[:5], sc.cpu().numpy()[bad[:5]], wsc[bad[:5]])
        assert np.array_equal(amd.u32(sk), wsk), sv
        # the same with 16-bit integer lanes in the DP instead of binary16 ones
        for kw in (dict(algo_flags=amd.ALN_NO_F16_DP),):
            batch = amd.AlignmentBatch(orc.pack4(flat), 4, roffs, orc.pack2(text), 2, wb, we, max_read_len=M, **kw)
            sc, sk = amd.batch_banded_alignment_score(31, amd.make_gotoh_aligner(oracle.SEMI_GLOBAL, _scheme(amd, sv)), batch)
            assert np.array_equal(sc.cpu().numpy(), wsc) and np.array_equal(amd.u32(sk), wsk), (sv, sorted(kw))
    # the indel reads really are cases where a gapped alignment beats a 2-3 mismatch diagonal
    assert ((wsc[1200:2700] > -18) & (wsc[1200:2700] <= -3)).mean() > 0.5
    # the same jobs with base qualities under quality-dependent mismatch penalties (nvBowtie's default ramp 2..6 and two others):
    # the first pass sums the penalties of its candidate diagonals row by row (QUAL), the chances argue with the smallest penalty
    _with_qualities(amd, orc, rng, flat, roffs, text, wb, we, M)


def _quality_sets(rng, n):
    """base qualities: uniform 0..63, Illumina's four bins, mostly-40 with low-quality tails, all equal"""
    tails = np.full(n, 40, dtype=np.uint8); low = rng.random(n) < 0.15; tails[low] = rng.integers(0, 30, int(low.sum()))
    return (rng.integers(0, 64, n, dtype=np.uint8), rng.choice(np.array([2, 12, 23, 37], dtype=np.uint8), n), tails,
            np.full(n, 17, dtype=np.uint8))


def _with_qualities(amd, orc, rng, flat, roffs, text, wb, we, M, schemes=((0, 2, 6, -8, -3, -8, -3), (0, 1, 3, -5, -1, -5, -1), (0, 3, 10, -9, -2, -9, -2))):
    rid = np.arange(len(wb), dtype=np.uint32)
    fl = np.zeros(len(wb), dtype=np.uint8)
    for quals in _quality_sets(rng, len(flat)):
        for sv in schemes:
            wsc, wsk = orc.banded_gotoh_packed_batch(31, oracle.SEMI_GLOBAL, oracle.Scheme(*sv), orc.pack4(flat), roffs, orc.pack2(text), wb, we,
                                                     read_id=rid, flags=fl, quals=quals)
            for algo in (None, amd.ALN_NO_UNGAPPED_SCORE, amd.ALN_NO_QUALITY_SHORTCUT, amd.ALN_NO_GAP_CHANCE):
                batch = amd.AlignmentBatch(orc.pack4(flat), 4, roffs, orc.pack2(text), 2, wb, we, quals=quals, max_read_len=M, algo_flags=algo)
                sc, sk = amd.batch_banded_alignment_score(31, amd.make_gotoh_aligner(oracle.SEMI_GLOBAL, _scheme(amd, sv)), batch)
                bad = np.nonzero(sc.cpu().numpy() != wsc)[0]
                assert len(bad) == 0, (sv, algo, bad[:5], sc.cpu().numpy()[bad[:5]], wsc[bad[:5]])
                assert np.array_equal(amd.u32(sk), wsk), (sv, algo)


def test_end_to_end_shortcut_with_qualities_reversed_reads(amd, orc):
    """the quality-aware first pass on reverse-complemented reads (quality bytes are taken in storage order, mirrored with the read),
    ragged read lengths and reads with N"""
    rng = np.random.default_rng(77)
    G, R = 300000, 5000
    text = rng.integers(0, 4, G, dtype=np.uint8)
    lens = rng.integers(60, 151, R)
    roffs = np.zeros(R + 1, dtype=np.uint32); roffs[1:] = np.cumsum(lens)
    starts = rng.integers(40, G - 300, R)
    rc = rng.random(R) < 0.5
    reads = []
    for k in range(R):
        r = text[starts[k]:starts[k] + lens[k]].copy()
        nm = int(rng.integers(0, 4)); pos = rng.choice(lens[k], nm, replace=False); r[pos] = (r[pos] + 1 + rng.integers(0, 3, nm)) % 4
        if rng.random() < 0.1 and lens[k] > 20:
            cpos = int(rng.integers(5, lens[k] - 5)); r = np.concatenate([r[:cpos], r[cpos + 1:], text[starts[k] + lens[k]:starts[k] + lens[k] + 1]])
        if rc[k]:
            r = (3 - r)[::-1]
        if rng.random() < 0.05:
            r = r.copy(); r[int(rng.integers(0, lens[k]))] = 4
        reads.append(r.astype(np.uint8))
    flat = np.concatenate(reads)
    wb = (starts - 15 + rng.integers(-2, 3, R)).astype(np.uint32); we = (wb + lens + 31).astype(np.uint32)
    flags = (rc * (amd.READ_REVERSE | amd.READ_COMPLEMENT)).astype(np.uint8)
    rid = np.arange(R, dtype=np.uint32)
    sv = (0, 2, 6, -8, -3, -8, -3)
    for quals in _quality_sets(rng, len(flat)):
        wsc, wsk = orc.banded_gotoh_packed_batch(31, oracle.SEMI_GLOBAL, oracle.Scheme(*sv), orc.pack4(flat), roffs, orc.pack2(text), wb, we,
                                                 read_id=rid, flags=flags, quals=quals)
        for algo in (None, amd.ALN_NO_UNGAPPED_SCORE, amd.ALN_RAGGED_READS, amd.ALN_RAGGED_READS | amd.ALN_NO_LENGTH_SORT,
                     amd.ALN_RAGGED_READS | amd.ALN_NO_UNGAPPED_SCORE, amd.ALN_RAGGED_READS | amd.ALN_NO_UNGAPPED_SCORE | amd.ALN_NO_LENGTH_SORT,
                     amd.ALN_RAGGED_READS | amd.ALN_NO_F16_DP):
            for kw in ({},):
                batch = amd.AlignmentBatch(orc.pack4(flat), 4, roffs, orc.pack2(text), 2, wb, we, quals=quals, flags=flags, max_read_len=150, algo_flags=algo, **kw)
                sc, sk = amd.batch_banded_alignment_score(31, amd.make_gotoh_aligner(oracle.SEMI_GLOBAL, _scheme(amd, sv)), batch)
                assert np.array_equal(sc.cpu().numpy(), wsc), (algo, bool(kw))
                assert np.array_equal(amd.u32(sk), wsk), (algo, bool(kw))
    assert (wsc > -8).mean() > 0.3                                    # a good share is settled by the first pass


@pytest.mark.parametrize("flags", [0, 1, 2 | 128, 1 | 32])       # 32: the packed kernel's 3-waves-per-SIMD build
def test_end_to_end_banded_scoring_on_low_complexity_text(amd, orc, flags, monkeypatch):
    """band-31 end-to-end scoring where equal scores are everywhere: tandem repeats of period 1-7 with a few mutations, two-letter
    stretches, periodic reads, 0-5 substitutions and an occasional indel -- many diagonals and many single- and double-gap alignments
    tie, which is what the shortcut's second and third chance have to get right (a tie goes to the DP) and what decides the sink.
    Default route, DP for every job (NVBIO_ALN_NO_UNGAPPED_SCORE) and first pass + DP only: scores and sinks equal the reference
    algorithm's."""
    monkeypatch.setattr(amd, "DEFAULT_ALGO_FLAGS", flags)
    rng = np.random.default_rng(515)
    R, M = 4000, 150
    G = R * 300
    text = rng.integers(0, 4, G, dtype=np.uint8)
    reads, wbs = [], []
    for j in range(R):
        base = j * 300 + 40
        kind = j % 3
        if kind == 0:
            unit = rng.integers(0, 4, int(rng.integers(1, 8))).astype(np.uint8)
            L = int(rng.integers(60, 230)); a0 = base - 30 + int(rng.integers(0, 60))
            rep = np.resize(unit, L).copy(); mut = rng.random(L) < 0.02; rep[mut] = rng.integers(0, 4, int(mut.sum()))
            text[a0:a0 + L] = rep
        elif kind == 1:
            L = int(rng.integers(40, 200)); a0 = base - 20 + int(rng.integers(0, 60))
            text[a0:a0 + L] = rng.integers(0, 2, L) * int(rng.integers(1, 4))
        src = text[base:base + M + 4]
        r = src[:M].copy()
        if rng.random() < 0.3:
            cpos = int(rng.integers(3, M - 3)); g = int(rng.integers(1, 4))
            r = np.concatenate([src[:cpos], src[cpos + g:]])[:M] if rng.random() < 0.5 else np.concatenate([src[:cpos], src[cpos - g:cpos], src[cpos:]])[:M]
        k = int(rng.integers(0, 6))
        if k:
            pos = rng.choice(M, k, replace=False); r[pos] = (r[pos] + 1 + rng.integers(0, 3, k)) % 4
        reads.append(r.astype(np.uint8)); wbs.append(base - 15 + int(rng.integers(-4, 5)))
    flat = np.concatenate(reads)
    roffs = (np.arange(R + 1) * M).astype(np.uint32)
    wb = np.array(wbs, dtype=np.uint32); we = (wb + 31 + M).astype(np.uint32)
    for sv in ((0, 6, 6, -8, -3, -8, -3), (0, 3, 3, -4, -2, -4, -2)):
        wsc, wsk = orc.banded_gotoh_packed_batch(31, oracle.SEMI_GLOBAL, oracle.Scheme(*sv), orc.pack4(flat), roffs, orc.pack2(text), wb, we)
        batch = amd.AlignmentBatch(orc.pack4(flat), 4, roffs, orc.pack2(text), 2, wb, we, max_read_len=M)
        sc, sk = amd.batch_banded_alignment_score(31, amd.make_gotoh_aligner(oracle.SEMI_GLOBAL, _scheme(amd, sv)), batch)
        bad = np.nonzero(sc.cpu().numpy() != wsc)[0]
        assert len(bad) == 0, (sv, bad[:5], sc.cpu().numpy()[bad[:5]], wsc[bad[:5]])
        assert np.array_equal(amd.u32(sk), wsk), sv
    assert (wsc <= -4).mean() > 0.5
    if flags in (0, 1):
        _with_qualities(amd, orc, rng, flat, roffs, text, wb, we, M, schemes=((0, 2, 6, -8, -3, -8, -3), (0, 1, 2, -4, -2, -4, -2)))


@pytest.mark.parametrize("typ", ["LOCAL", "SEMI_GLOBAL"])
def test_jobs_longer_than_the_declared_bound_are_rejected(amd, orc, typ):
    """nvbio_alignment_batch::max_read_len is what lets the library pick 16-bit kernels: a job whose pattern is longer than a
    non-zero bound is rejected (score NVBIO_SCORE_MIN, sink (-1,-1)) by every kernel alike instead of being scored in registers
    that could wrap; the other jobs are unaffected"""
    typ = getattr(oracle, typ)
    rng = np.random.default_rng(90)
    G, R = 100000, 600
    text = rng.integers(0, 4, G, dtype=np.uint8)
    lens = np.where(np.arange(R) % 7 == 0, 150, 100)                    # every 7th read is longer than the bound declared below
    roffs = np.zeros(R + 1, dtype=np.uint32); roffs[1:] = np.cumsum(lens)
    starts = rng.integers(20, G - 200, R)
    reads = np.concatenate([text[s:s + l] for s, l in zip(starts, lens)]).copy()
    reads[rng.random(len(reads)) < 0.02] = rng.integers(0, 4)
    wb = (starts - 15).astype(np.uint32); we = (wb + lens + 31).astype(np.uint32)
    scheme = amd.GotohScheme(2, 6, 6, -8, -3, -8, -3) if typ == oracle.LOCAL else amd.GotohScheme(0, 6, 6, -8, -3, -8, -3)
    osc = oracle.Scheme(*[int(getattr(scheme.c, f)) for f, _ in scheme.c._fields_])
    want_s, want_k = orc.banded_gotoh_packed_batch(31, typ, osc, orc.pack4(reads), roffs, orc.pack2(text), wb, we)
    long_ = lens > 100
    for flags in (0, amd.ALN_NO_PACKED_DP, amd.ALN_NO_UNGAPPED_SCORE):
        batch = amd.AlignmentBatch(orc.pack4(reads), 4, roffs, orc.pack2(text), 2, wb, we, max_read_len=100, algo_flags=flags)
        sc, sk = amd.batch_banded_alignment_score(31, amd.make_gotoh_aligner(typ, scheme), batch)
        sc, sk = sc.cpu().numpy(), amd.u32(sk)
        assert (sc[long_] == amd.SCORE_MIN).all() and (sk[long_] == 0xFFFFFFFF).all()
        assert np.array_equal(sc[~long_], want_s[~long_]) and np.array_equal(sk[~long_], want_k[~long_])
    # with an honest bound (or none) every job is scored
    for mrl in (150, 0):
        batch = amd.AlignmentBatch(orc.pack4(reads), 4, roffs, orc.pack2(text), 2, wb, we, max_read_len=mrl)
        sc, sk = amd.batch_banded_alignment_score(31, amd.make_gotoh_aligner(typ, scheme), batch)
        assert np.array_equal(sc.cpu().numpy(), want_s) and np.array_equal(amd.u32(sk), want_k)


def test_gap_chance_on_reads_with_indels(amd, orc):
    """the gap chance (one-gap alignments of a job without a near-clean diagonal, evaluated exactly): reads with ONE indel of 1-7 symbols
    anywhere and 0-3 substitutions, reads with TWO indels of 1-2 symbols (the two-gap classes it must rule out or hand to the DP), indels
    inside homopolymers and short tandem repeats (several placements and end columns tie), reads whose best alignment is worse than anything
    it evaluates -- several schemes (which classes lie below the cheapest unknown one depends on the numbers), with and without the chance:
    scores and sinks equal the reference algorithm's"""
    rng = np.random.default_rng(4242)
    G = 600000
    text = rng.integers(0, 4, G, dtype=np.uint8)
    runs = rng.integers(200, G - 400, 400)                          # homopolymers and period-2/3 runs sprinkled over the text
    for p0 in runs:
        L = int(rng.integers(5, 30)); unit = rng.integers(0, 4, int(rng.integers(1, 4))).astype(np.uint8)
        text[p0:p0 + L] = np.resize(unit, L)
    M = 150
    reads, wbs = [], []

    def mutate(r, k):
        if k:
            pos = rng.choice(len(r), k, replace=False); r[pos] = (r[pos] + 1 + rng.integers(0, 3, k)) % 4
        return r

    def with_indel(src, at, g, ins):
        return np.concatenate([src[:at], rng.integers(0, 4, g, dtype=np.uint8), src[at:]]) if ins else np.concatenate([src[:at], src[at + g:]])

    for j in range(6000):
        near_run = j % 3 == 0
        p0 = int(runs[j % len(runs)] - rng.integers(20, 120)) if near_run else int(rng.integers(100, G - 400))
        src = text[p0:p0 + M + 16].copy()
        kind = j % 10
        if kind < 7:                                                # one indel
            g = int(rng.integers(1, 8)); at = int(rng.integers(1, M - 8))
            r = with_indel(src, at, g, rng.random() < 0.5)[:M]
            r = mutate(r.copy(), int(rng.integers(0, 4)))
        elif kind < 9:                                              # two indels
            a1, a2 = sorted(rng.integers(2, M - 6, 2)); g1, g2 = int(rng.integers(1, 3)), int(rng.integers(1, 3))
            r = with_indel(with_indel(src, int(a2), g2, rng.random() < 0.5), int(a1), g1, rng.random() < 0.5)[:M]
            r = mutate(r.copy(), int(rng.integers(0, 2)))
        else:                                                       # many substitutions, no indel
            r = mutate(src[:M].copy(), int(rng.integers(5, 12)))
        reads.append(r.astype(np.uint8)); wbs.append(p0 - 15 + int(rng.integers(-6, 7)))
    R = len(reads)
    flat = np.concatenate(reads)
    roffs = (np.arange(R + 1) * M).astype(np.uint32)
    wb = np.array(wbs, dtype=np.uint32); we = (wb + 31 + M).astype(np.uint32)
    for sv in ((0, 6, 6, -8, -3, -8, -3), (0, 2, 2, -5, -1, -5, -1), (0, 4, 4, -6, -6, -6, -6), (0, 3, 3, -4, -2, -4, -2), (0, 6, 6, -5, -3, -5, -3),
               (0, 9, 9, -4, -2, -4, -2), (0, 1, 1, -7, -1, -7, -1), (0, 5, 5, -2, -2, -2, -2)):
        wsc, wsk = orc.banded_gotoh_packed_batch(31, oracle.SEMI_GLOBAL, oracle.Scheme(*sv), orc.pack4(flat), roffs, orc.pack2(text), wb, we)
        for algo in (None, amd.ALN_NO_GAP_CHANCE):
            batch = amd.AlignmentBatch(orc.pack4(flat), 4, roffs, orc.pack2(text), 2, wb, we, max_read_len=M, algo_flags=algo)
            sc, sk = amd.batch_banded_alignment_score(31, amd.make_gotoh_aligner(oracle.SEMI_GLOBAL, _scheme(amd, sv)), batch)
            bad = np.nonzero((sc.cpu().numpy() != wsc) | (amd.u32(sk) != wsk).any(axis=1))[0]
            assert len(bad) == 0, (sv, algo, bad[:5], sc.cpu().numpy()[bad[:5]], wsc[bad[:5]], amd.u32(sk)[bad[:5]], wsk[bad[:5]])
    assert ((wsc < -8) & (wsc > -60)).mean() > 0.5
    # ... and under quality ramps: members are priced with the penalties of the rows that really mismatch, every split of them between prefix and suffix
    _with_qualities(amd, orc, rng, flat, roffs, text, wb, we, M, schemes=((0, 2, 6, -8, -3, -8, -3), (0, 1, 3, -5, -1, -5, -1), (0, 3, 10, -9, -2, -9, -2), (0, 4, 5, -6, -3, -6, -3)))


def test_gap_chance_on_ragged_reversed_reads(amd, orc):
    """the gap chance where nothing is uniform: read lengths 33..161 (and a few shorter than a plane word), reads stored reversed and / or
    complemented, N symbols, windows shifted and clipped -- one indel of 1-5 symbols or two small ones or many substitutions per read; with the
    ragged-batch hint and without, with and without the chance"""
    rng = np.random.default_rng(777)
    G, R = 500000, 7000
    text = rng.integers(0, 4, G, dtype=np.uint8)
    lens = rng.integers(33, 162, R); lens[:60] = rng.integers(8, 33, 60)
    roffs = np.zeros(R + 1, dtype=np.uint32); roffs[1:] = np.cumsum(lens)
    starts = rng.integers(60, G - 400, R)
    flags = rng.integers(0, 4, R).astype(np.uint8)
    reads = []
    for k in range(R):
        L = int(lens[k]); src = text[starts[k]:starts[k] + L + 12].copy()
        kind = k % 5
        if kind < 3 and L > 12:
            g = int(rng.integers(1, 6)); at = int(rng.integers(1, L - 6))
            r = (np.concatenate([src[:at], rng.integers(0, 4, g, dtype=np.uint8), src[at:]]) if rng.random() < 0.5 else np.concatenate([src[:at], src[at + g:]]))[:L]
            nm = int(rng.integers(0, 3))
        elif kind == 3 and L > 20:
            a1, a2 = sorted(rng.integers(2, L - 6, 2))
            r = np.concatenate([src[:a1], src[a1 + 1:a2], rng.integers(0, 4, 1, dtype=np.uint8), src[a2:]])[:L]
            nm = int(rng.integers(0, 2))
        else:
            r = src[:L]; nm = int(rng.integers(3, 9))
        r = r.copy()
        if nm:
            pos = rng.choice(L, min(nm, L), replace=False); r[pos] = (r[pos] + 1 + rng.integers(0, 3, len(pos))) % 4
        if rng.random() < 0.1:
            r[int(rng.integers(0, L))] = 4
        # store the read so that the job's flags give back `r` as aligned: reverse / complement are involutions
        if flags[k] & 2:
            r = np.where(r < 4, 3 - r, r)
        if flags[k] & 1:
            r = r[::-1]
        reads.append(r.astype(np.uint8))
    flat = np.concatenate(reads)
    wb = (starts - 15 + rng.integers(-4, 5, R)).astype(np.uint32); we = (wb + lens + 31).astype(np.uint32)
    we[::23] -= rng.integers(1, 12, len(we[::23])).astype(np.uint32)           # clipped windows: not the chance's
    rid = np.arange(R, dtype=np.uint32)
    for sv in ((0, 6, 6, -8, -3, -8, -3), (0, 3, 3, -4, -2, -4, -2)):
        wsc, wsk = orc.banded_gotoh_packed_batch(31, oracle.SEMI_GLOBAL, oracle.Scheme(*sv), orc.pack4(flat), roffs, orc.pack2(text), wb, we, read_id=rid, flags=flags)
        for algo in (None, amd.ALN_RAGGED_READS, amd.ALN_NO_GAP_CHANCE, amd.ALN_RAGGED_READS | amd.ALN_NO_GAP_CHANCE | amd.ALN_NO_F16_DP):
            batch = amd.AlignmentBatch(orc.pack4(flat), 4, roffs, orc.pack2(text), 2, wb, we, flags=flags, max_read_len=161, algo_flags=algo)
            sc, sk = amd.batch_banded_alignment_score(31, amd.make_gotoh_aligner(oracle.SEMI_GLOBAL, _scheme(amd, sv)), batch)
            bad = np.nonzero((sc.cpu().numpy() != wsc) | (amd.u32(sk) != wsk).any(axis=1))[0]
            assert len(bad) == 0, (sv, algo, bad[:5], sc.cpu().numpy()[bad[:5]], wsc[bad[:5]], amd.u32(sk)[bad[:5]], wsk[bad[:5]], lens[bad[:5]], flags[bad[:5]])
