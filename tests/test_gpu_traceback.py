"""GPU parity: batched banded Gotoh traceback (score, source, sink, run-length CIGAR) through the C-ABI
vs the reference's golden vectors (tests/golden/tb_golden.npz) and the oracle on nvBowtie-shaped batches."""
import numpy as np
import pytest

import oracle

pytestmark = pytest.mark.gpu


def _scheme(amd, v):
    return amd.GotohScheme(*[int(x) for x in v])


def _i64(t):
    return t.cpu().numpy().astype(np.int64)


@pytest.fixture(params=["shortcut", "dp-only", "no-narrow", "no-band-route"])
def tb_mode(request, monkeypatch, amd):
    """the default route (ungapped shortcut; band-15 DP for the end-to-end jobs whose optimal paths stay within 7 diagonals of the
    sink), the plain DP-for-every-job path, and the shortcut with every DP over the whole band must all equal the reference"""
    if request.param == "dp-only":
        monkeypatch.setattr(amd, "DEFAULT_ALGO_FLAGS", amd.ALN_NO_UNGAPPED_TRACEBACK)
    if request.param == "no-narrow":
        monkeypatch.setattr(amd, "DEFAULT_ALGO_FLAGS", amd.ALN_NO_NARROW_TRACEBACK)
    if request.param == "no-band-route":                            # full-matrix traceback: restricted rows instead of the band-15 kernel
        monkeypatch.setattr(amd, "DEFAULT_ALGO_FLAGS", amd.ALN_NO_BAND_ROUTE)
    return request.param


def test_traceback_golden(amd, dp_golden, tb_golden, tb_mode):
    g, t = dp_golden, tb_golden
    S = len(g["schemes"])
    nk = int(g["n_known"])
    n = len(g["pat_off"]) - 1
    max_len = int(np.diff(g["pat_off"]).max())
    groups = {}
    for i in range(n):
        key = ("k", i) if i < nk else ("s", i % S, int(g["has_quals"][i]))
        groups.setdefault(key, []).append(i)
    checked = 0
    STRIDE = 96
    for key, cases in groups.items():
        cases = np.array(cases, dtype=np.uint32)
        sv = g["known_schemes"][cases[0]] if key[0] == "k" else g["schemes"][key[1]]
        hq = bool(g["has_quals"][cases[0]])
        batch = amd.AlignmentBatch(g["pats"], 8, g["pat_off"], g["txts"], 8, g["txt_off"][cases], g["txt_off"][cases + 1],
                                   quals=g["quals"] if hq else None, read_id=cases, max_read_len=max_len)
        for bi, band in enumerate(t["bands"]):
            for typ in range(3):
                al = amd.make_gotoh_aligner(typ, _scheme(amd, sv))
                sc, src, snk, cig, ln = amd.BatchedBandedAlignmentTraceback(int(band), al).enact(batch, cigar_stride=STRIDE)
                sc, src, snk, ln = _i64(sc), _i64(src), _i64(snk), _i64(ln)
                cig = cig.cpu().numpy().view(np.uint16)
                for k, i in enumerate(cases):
                    want = t["aln"][i, bi, typ]
                    if want[0] < 0:
                        continue                                   # the reference reads past a < BAND-1 text: not pinned
                    lo, hi = t["cig_off"][i, bi, typ]
                    assert sc[k] == want[1], (i, band, typ)
                    assert tuple(src[k]) == (want[2], want[3]) and tuple(snk[k]) == (want[4], want[5]), (i, band, typ)
                    assert ln[k] == hi - lo, (i, band, typ)
                    m = min(int(ln[k]), STRIDE)                 # elements beyond the stride are dropped, the count is not
                    assert np.array_equal(cig[k, :m], t["cigars"][lo:lo + m]), (i, band, typ)
                    checked += 1
    assert checked > 4000
    # the reference's functional tests (alignment_test.cu:743-746,779-785): 4M1D3M and 147M2D3M in backtracking order
    for k, band, sv, want in ((0, 7, (2, 1, 1, -1, -1, -1, -1), "4M1D3M"), (1, 31, (0, 5, 5, -8, -3, -8, -3), "147M2D3M")):
        c = np.array([k], dtype=np.uint32)
        batch = amd.AlignmentBatch(g["pats"], 8, g["pat_off"], g["txts"], 8, g["txt_off"][c], g["txt_off"][c + 1], read_id=c,
                                   max_read_len=max_len)
        out = amd.BatchedBandedAlignmentTraceback(band, amd.make_gotoh_aligner(oracle.SEMI_GLOBAL, _scheme(amd, sv))).enact(batch)
        assert amd.cigar_string(out[3][0].cpu().numpy(), int(out[4][0]), forward=False) == want


@pytest.mark.parametrize("typ", ["LOCAL", "SEMI_GLOBAL"])
def test_traceback_packed_batch_vs_oracle(amd, orc, typ, tb_mode):
    """nvBowtie-shaped: 4-bit reads with reversed / complemented flags and qualities against 2-bit genome
    windows (clipped at both genome ends, some shorter than the read), ragged read lengths, chunked scratch"""
    import torch
    typ = getattr(oracle, typ)
    rng = np.random.default_rng(5)
    G = 300000
    text = rng.integers(0, 4, G, dtype=np.uint8)
    R = 3000
    lens = rng.integers(30, 151, R); lens[::2] = 150
    roffs = np.zeros(R + 1, dtype=np.uint32); roffs[1:] = np.cumsum(lens)
    starts = rng.integers(0, G - 200, R)
    reads = []
    for s, l in zip(starts, lens):
        r = text[s:s + l].copy()
        mut = rng.random(l) < 0.02
        r[mut] = rng.integers(0, 4, int(mut.sum()))
        if rng.random() < 0.4:                                    # an indel of 1-3 bp
            k = int(rng.integers(5, l - 5)); gsz = int(rng.integers(1, 4))
            r = np.concatenate([r[:k], r[k + gsz:], rng.integers(0, 4, gsz, dtype=np.uint8)]) if rng.random() < 0.5 \
                else np.concatenate([r[:k], rng.integers(0, 4, gsz, dtype=np.uint8), r[k:l - gsz]])
        reads.append(r)
    flat = np.concatenate(reads)
    flat[rng.integers(0, len(flat), 100)] = 4
    quals = rng.integers(0, 64, len(flat), dtype=np.uint8)
    J = 9001
    rid = rng.integers(0, R, J).astype(np.uint32)
    flags = np.zeros(J, dtype=np.uint8); flags[J // 2:] = rng.integers(0, 4, J - J // 2)
    g_pos = starts[rid].astype(np.int64) + rng.integers(-4, 5, J)
    g_pos[::50] = rng.integers(0, 10, len(g_pos[::50]))
    g_pos[1::50] = G - rng.integers(20, 170, len(g_pos[1::50]))
    g_pos = np.clip(g_pos, 0, G - 1)
    wb = np.where(g_pos > 15, g_pos - 15, 0).astype(np.uint32)
    we = np.minimum(wb + 31 + lens[rid], G).astype(np.uint32)
    sel = np.nonzero((we - wb) >= 30)[0]
    stride = 48
    for sv in ((2, 2, 6, -8, -3, -8, -3), (0, 6, 6, -8, -3, -8, -3), (1, 3, 3, -11, -4, -6, -2)):
        for use_q in (True, False):
            kw = dict(quals=quals if use_q else None, read_id=rid[sel], flags=flags[sel])
            want = orc.banded_gotoh_traceback_packed_batch(31, typ, oracle.Scheme(*sv), orc.pack4(flat), roffs, orc.pack2(text),
                                                           wb[sel], we[sel], stride, **kw)
            batch = amd.AlignmentBatch(orc.pack4(flat), 4, roffs, orc.pack2(text), 2, wb[sel], we[sel], max_read_len=150, **kw)
            op = amd.BatchedBandedAlignmentTraceback(31, amd.make_gotoh_aligner(typ, _scheme(amd, sv)))
            need = op.min_temp_storage(batch)
            assert need == len(sel) * 150 * 16
            # caller scratch for a third of the jobs: the batch is processed in several launches
            temp = torch.empty(need // 3 + 16, dtype=torch.uint8, device="cuda:0")
            for tmp in (None, temp):
                sc, src, snk, cig, ln = op.enact(batch, cigar_stride=stride, temp=tmp)
                assert np.array_equal(sc.cpu().numpy(), want[0]), (sv, use_q)
                assert np.array_equal(amd.u32(src), want[1]) and np.array_equal(amd.u32(snk), want[2]), (sv, use_q)
                assert np.array_equal(amd.u32(ln), want[4]), (sv, use_q)
                assert np.array_equal(cig.cpu().numpy().view(np.uint16), want[3]), (sv, use_q)
            # the scores and sinks are those of the scoring kernel
            s2, k2 = amd.batch_banded_alignment_score(31, amd.make_gotoh_aligner(typ, _scheme(amd, sv)), batch)
            assert np.array_equal(s2.cpu().numpy(), want[0]) and np.array_equal(amd.u32(k2), want[2])
            # ... and can be handed over instead of being recomputed (NVBIO_TRACEBACK_SINKS_GIVEN)
            sc, src, snk, cig, ln = op.enact(batch, cigar_stride=stride, scores=s2, sinks=k2)
            assert np.array_equal(sc.cpu().numpy(), want[0]) and np.array_equal(amd.u32(snk), want[2])
            assert np.array_equal(amd.u32(src), want[1]) and np.array_equal(amd.u32(ln), want[4])
            assert np.array_equal(cig.cpu().numpy().view(np.uint16), want[3]), (sv, use_q)
    assert (want[4] > 3).any() and (want[4] == 0).any()           # gapped alignments and untraceable jobs are both in
    # finish_alignment on the last traceback: edit distance and MDS stream of every job (nvBowtie traceback_inl.h:536-705)
    ed, mds, ml = amd.finish_alignment(batch, src, cig, ln, mds_stride=96)
    ed, mds, ml = amd.u32(ed), mds.cpu().numpy(), amd.u32(ml)
    lens_h = amd.u32(ln); src_h = amd.u32(src); cig_h = cig.cpu().numpy().view(np.uint16); sc_h = sc.cpu().numpy()
    checked = 0
    for k, j in enumerate(sel):
        if lens_h[k] == 0:
            assert ed[k] == 0 and ml[k] == 0
            continue
        r = rid[j]; p = flat[roffs[r]:roffs[r + 1]]
        if flags[j] & 1:
            p = p[::-1]
        if flags[j] & 2:
            p = np.where(p < 4, 3 - p, p).astype(np.uint8)
        wed, wmds = orc.finish_alignment(p, text[wb[j]:we[j]], cig_h[k, :lens_h[k]], int(src_h[k, 0]))
        assert ed[k] == wed and ml[k] == len(wmds), (k, j)
        assert np.array_equal(mds[k, :min(ml[k], 96)], wmds[:96]), (k, j)
        checked += 1
        if typ == oracle.SEMI_GLOBAL and not use_q and sv == (1, 3, 3, -11, -4, -6, -2) and lens_h[k] == 1:
            # an ungapped end-to-end alignment: score = matches - 3 mismatches
            assert sc_h[k] == (len(p) - wed) * 1 - 3 * wed
    assert checked > 5000


def _low_complexity_jobs(rng, R, M, span):
    """windows of `span` symbols full of tandem repeats (period 1-7, 2 % mutated), two-letter stretches and plain random text; reads
    taken from inside them with 0-4 substitutions and sometimes an indel of 1-3 symbols: co-optimal alignments everywhere"""
    G = R * (span + 60)
    text = rng.integers(0, 4, G, dtype=np.uint8)
    reads, loci = [], []
    for j in range(R):
        base = j * (span + 60) + 30
        kind = j % 3
        if kind == 0:
            unit = rng.integers(0, 4, int(rng.integers(1, 8))).astype(np.uint8)
            L = int(rng.integers(M // 2, span)); a0 = base + int(rng.integers(0, span - L + 1))
            rep = np.resize(unit, L).copy(); mut = rng.random(L) < 0.02; rep[mut] = rng.integers(0, 4, int(mut.sum()))
            text[a0:a0 + L] = rep
        elif kind == 1:
            L = int(rng.integers(M // 3, span)); a0 = base + int(rng.integers(0, span - L + 1))
            text[a0:a0 + L] = rng.integers(0, 2, L) * int(rng.integers(1, 4))
        a = base + int(rng.integers(0, span - M - 4))
        src = text[a:a + M + 4]
        r = src[:M].copy()
        if rng.random() < 0.4:
            cpos = int(rng.integers(3, M - 3)); g = int(rng.integers(1, 4))
            r = np.concatenate([src[:cpos], src[cpos + g:]])[:M] if rng.random() < 0.5 else np.concatenate([src[:cpos], src[cpos - g:cpos], src[cpos:]])[:M]
        k = int(rng.integers(0, 5))
        if k:
            pos = rng.choice(M, k, replace=False); r[pos] = (r[pos] + 1 + rng.integers(0, 3, k)) % 4
        reads.append(r.astype(np.uint8)); loci.append(a)
    return text, reads, np.array(loci, dtype=np.int64)


def test_traceback_on_low_complexity_text(amd, orc, tb_mode):
    """banded and full-matrix end-to-end tracebacks where co-optimal alignments abound (tandem repeats, two-letter stretches): the
    narrow-band / restricted-row routes rest on every co-optimal path lying inside the region they compute, so that the reference's
    tie rules meet the same values -- scores, sources, sinks and CIGARs equal the reference algorithm's in every mode"""
    rng = np.random.default_rng(717)
    sv = (0, 6, 6, -8, -3, -8, -3)
    stride = 40
    # banded: a window of M + 31 around the locus
    R, M = 3000, 150
    text, reads, loci = _low_complexity_jobs(rng, R, M, 220)
    flat = np.concatenate(reads); roffs = (np.arange(R + 1) * M).astype(np.uint32)
    wb = (loci - 15 + rng.integers(-4, 5, R)).astype(np.uint32); we = (wb + 31 + M).astype(np.uint32)
    want = orc.banded_gotoh_traceback_packed_batch(31, oracle.SEMI_GLOBAL, oracle.Scheme(*sv), orc.pack4(flat), roffs, orc.pack2(text), wb, we, stride)
    batch = amd.AlignmentBatch(orc.pack4(flat), 4, roffs, orc.pack2(text), 2, wb, we, max_read_len=M)
    sc, src, snk, cig, ln = amd.BatchedBandedAlignmentTraceback(31, amd.make_gotoh_aligner(oracle.SEMI_GLOBAL, _scheme(amd, sv))).enact(batch, cigar_stride=stride)
    assert np.array_equal(sc.cpu().numpy(), want[0]) and np.array_equal(amd.u32(src), want[1]) and np.array_equal(amd.u32(snk), want[2])
    assert np.array_equal(amd.u32(ln), want[4]) and np.array_equal(cig.cpu().numpy().view(np.uint16), want[3])
    assert (want[4] > 2).mean() > 0.2
    # full matrix: the opposite-mate shape, 150 in 400
    R, M, W = 1200, 150, 400
    text, reads, loci = _low_complexity_jobs(rng, R, M, W)
    flat = np.concatenate(reads); roffs = (np.arange(R + 1) * M).astype(np.uint32)
    wb = (np.arange(R) * (W + 60) + 30).astype(np.uint32); we = (wb + W).astype(np.uint32)
    batch = amd.AlignmentBatch(orc.pack4(flat), 4, roffs, orc.pack2(text), 2, wb, we)
    sc, src, snk, cig, ln = amd.BatchedAlignmentTraceback(amd.make_gotoh_aligner(oracle.SEMI_GLOBAL, _scheme(amd, sv))).enact(batch, M, W, cigar_stride=stride)
    sc, src, snk, ln = sc.cpu().numpy(), amd.u32(src), amd.u32(snk), amd.u32(ln)
    cig = cig.cpu().numpy().view(np.uint16)
    gapped = 0
    for j in range(R):
        ok, s_, so_, sk_, c_ = orc.full_gotoh_traceback(oracle.SEMI_GLOBAL, oracle.Scheme(*sv), reads[j], text[wb[j]:we[j]])
        assert sc[j] == s_ and tuple(src[j]) == so_ and tuple(snk[j]) == sk_, j
        assert ln[j] == len(c_) and np.array_equal(cig[j, :min(len(c_), stride)], c_[:stride]), j
        gapped += len(c_) > 2
    assert gapped > 200


def test_sw_traceback_golden(amd, swtb_golden, tb_mode):
    """nvbio_banded_sw_traceback (the linear-gap Smith-Waterman / edit-distance aligners through BatchedBandedAlignmentTraceback) against
    the reference's own outputs: scores, sources, sinks, CIGARs; unequal deletion / insertion costs; LOCAL walks that run to row 0"""
    g = swtb_golden
    max_len = int(np.diff(g["pat_off"]).max())
    STRIDE = 64
    seen = 0
    for band in (3, 7, 15, 31):
        for typ in range(3):
            for si in range(len(g["schemes"])):
                sel = np.nonzero((g["band"] == band) & (g["typ"] == typ) & (g["scheme"] == si))[0].astype(np.uint32)
                if len(sel) == 0:
                    continue
                batch = amd.AlignmentBatch(g["pats"], 8, g["pat_off"], g["txts"], 8, g["txt_off"][sel], g["txt_off"][sel + 1], read_id=sel,
                                           max_read_len=max_len)
                al = amd.make_smith_waterman_aligner(typ, amd.SimpleSmithWatermanScheme(*[int(v) for v in g["schemes"][si]]))
                sc, src, snk, cig, ln = amd.BatchedBandedAlignmentTraceback(band, al).enact(batch, cigar_stride=STRIDE)
                sc, src, snk, ln = _i64(sc), amd.u32(src).astype(np.int64), amd.u32(snk).astype(np.int64), _i64(ln)
                cig = cig.cpu().numpy().view(np.uint16)
                for k, i in enumerate(sel):
                    want = g["out"][i]
                    assert sc[k] == want[1], (i, band, typ, si)
                    if want[0]:
                        lo, hi = int(g["cig_off"][i]), int(g["cig_off"][i + 1])
                        assert tuple(src[k]) == (want[2], want[3]) and tuple(snk[k]) == (want[4], want[5]), (i, band, typ, si)
                        assert ln[k] == hi - lo and np.array_equal(cig[k, :min(hi - lo, STRIDE)], g["cigars"][lo:lo + min(hi - lo, STRIDE)]), (i, band, typ, si)
                    else:
                        assert ln[k] == 0
                seen += len(sel)
    assert seen == len(g["band"])


def test_traceback_argument_errors(amd, orc):
    txt = np.zeros(256, dtype=np.uint8)
    pats = np.zeros(80, dtype=np.uint8)
    roffs = np.array([0, 40, 80], dtype=np.uint32)
    wb = np.array([0, 0], dtype=np.uint32); we = np.array([71, 64], dtype=np.uint32)
    al = amd.make_gotoh_aligner(oracle.LOCAL, amd.SimpleGotohScheme(2, -1, -2, -1))
    with pytest.raises(amd.NvbioError):                            # max_read_len is required
        amd.BatchedBandedAlignmentTraceback(31, al).enact(amd.AlignmentBatch(pats, 8, roffs, txt, 8, wb, we))
    with pytest.raises(amd.NvbioError):                            # unsupported band
        amd.BatchedBandedAlignmentTraceback(9, al).enact(amd.AlignmentBatch(pats, 8, roffs, txt, 8, wb, we, max_read_len=40))
    with pytest.raises(amd.NvbioError):                            # scores could overflow the int16 checkpoints
        amd.BatchedBandedAlignmentTraceback(31, amd.make_gotoh_aligner(oracle.LOCAL, amd.SimpleGotohScheme(900, -1, -2, -1))).enact(
            amd.AlignmentBatch(pats, 8, roffs, txt, 8, wb, we, max_read_len=40))
    # a pattern longer than max_read_len is skipped and flagged, the other job is traced: 40 matches
    out = amd.BatchedBandedAlignmentTraceback(31, al).enact(
        amd.AlignmentBatch(pats, 8, np.array([0, 30, 80], dtype=np.uint32), txt, 8, wb, np.array([61, 90], dtype=np.uint32), max_read_len=30))
    ln = amd.u32(out[4])
    assert ln[1] == 0xFFFFFFFF and ln[0] == 1 and int(out[0][0]) == 60
    assert amd.cigar_string(out[3][0].cpu().numpy(), 1) == "30M"


def test_full_traceback_edge_cases(amd, orc):
    """full-matrix tracebacks (Gotoh and Smith-Waterman aligners): an empty batch is a no-op; a text shorter than the pattern is traced
    (the full matrix has no band: the read's overhang becomes insertions / clips); a job beyond max_pattern_len / max_text_len is
    skipped and flagged; scores that could leave the reference's int16 checkpoints are refused"""
    rng = np.random.default_rng(5)
    txt = rng.integers(0, 4, 600, dtype=np.uint8)
    pats = np.concatenate([txt[100:140], txt[300:360], txt[10:30]]).astype(np.uint8)
    roffs = np.array([0, 40, 100, 120], dtype=np.uint32)
    wb = np.array([90, 290, 15], dtype=np.uint32); we = np.array([160, 380, 25], dtype=np.uint32)       # job 2: 10 text symbols for 20
    for al, ofn in ((amd.make_gotoh_aligner(oracle.SEMI_GLOBAL, amd.SimpleGotohScheme(0, -6, -8, -3)),
                     lambda p, t: orc.full_gotoh_traceback(oracle.SEMI_GLOBAL, oracle.Scheme(0, 6, 6, -8, -3, -8, -3), p, t)),
                    (amd.make_smith_waterman_aligner(oracle.SEMI_GLOBAL, amd.SimpleSmithWatermanScheme(0, -3, -5, -2)),
                     lambda p, t: orc.full_sw_traceback(oracle.SEMI_GLOBAL, (0, -3, -5, -2), p, t))):
        tb = amd.BatchedAlignmentTraceback(al)
        # empty batch
        e = amd.AlignmentBatch(pats, 8, roffs, txt, 8, wb[:0], we[:0], read_id=np.zeros(0, np.uint32))
        out = tb.enact(e, 60, 90)
        assert out[0].numel() == 0 and out[4].numel() == 0
        batch = amd.AlignmentBatch(pats, 8, roffs, txt, 8, wb, we)
        sc, src, snk, cig, ln = tb.enact(batch, 60, 90, cigar_stride=32)
        for j in range(3):
            p_, t_ = pats[roffs[j]:roffs[j + 1]], txt[wb[j]:we[j]]
            ok, s_, so_, sk_, c_ = ofn(p_, t_)
            assert int(sc[j]) == s_ and tuple(amd.u32(src)[j]) == so_ and tuple(amd.u32(snk)[j]) == sk_, j
            assert int(ln[j]) == len(c_) and np.array_equal(cig[j].cpu().numpy().view(np.uint16)[:len(c_)], c_), j
        assert int(sc[0]) == 0 and amd.cigar_string(cig[0].cpu().numpy(), int(ln[0])) == "40M"
        # a job beyond the declared bounds is skipped and flagged, the others are traced
        sc, src, snk, cig, ln = tb.enact(batch, 40, 90, cigar_stride=32)
        assert amd.u32(ln)[1] == 0xFFFFFFFF and amd.u32(ln)[0] == 1 and int(sc[0]) == 0
    with pytest.raises(amd.NvbioError):                            # scores could overflow the int16 checkpoints
        amd.BatchedAlignmentTraceback(amd.make_smith_waterman_aligner(oracle.LOCAL, amd.SimpleSmithWatermanScheme(300, -300, -300, -300))).enact(batch, 60, 90)


def test_full_traceback_golden(amd, dp_golden, ftb_golden, tb_mode):
    """full-matrix traceback through the C-ABI vs the reference's alignment_traceback (ftb_golden.npz): every pair,
    3 types, with and without min_score, shortcut on and off"""
    g, t = dp_golden, ftb_golden
    S = len(g["schemes"]); nk = int(g["n_known"])
    n = len(g["pat_off"]) - 1
    max_p = int(np.diff(g["pat_off"]).max()); max_t = int(np.diff(g["txt_off"]).max())
    groups = {}
    for i in range(n):
        key = ("k", i) if i < nk else ("s", i % S, int(g["has_quals"][i]))
        groups.setdefault(key, []).append(i)
    STRIDE = 160
    checked = traced = 0
    for key, cases in groups.items():
        cases = np.array(cases, dtype=np.uint32)
        sv = g["known_schemes"][cases[0]] if key[0] == "k" else g["schemes"][key[1]]
        hq = bool(g["has_quals"][cases[0]])
        batch = amd.AlignmentBatch(g["pats"], 8, g["pat_off"], g["txts"], 8, g["txt_off"][cases], g["txt_off"][cases + 1],
                                   quals=g["quals"] if hq else None, read_id=cases)
        for typ in range(3):
            for v in (0, 1):
                ms = g["min_scores"][cases] if v else None
                sc, src, snk, cig, ln = amd.BatchedAlignmentTraceback(amd.make_gotoh_aligner(typ, _scheme(amd, sv))).enact(
                    batch, max_p, max_t, min_scores=ms, cigar_stride=STRIDE)
                sc, src, snk, ln = _i64(sc), _i64(src), _i64(snk), _i64(ln)
                cig = cig.cpu().numpy().view(np.uint16)
                for k, i in enumerate(cases):
                    want = t["aln"][i, typ, v]
                    lo, hi = t["cig_off"][i, typ, v]
                    assert sc[k] == want[1], (i, typ, v)
                    assert tuple(src[k]) == (want[2], want[3]) and tuple(snk[k]) == (want[4], want[5]), (i, typ, v)
                    assert ln[k] == hi - lo, (i, typ, v)
                    m = min(int(ln[k]), STRIDE)
                    assert np.array_equal(cig[k, :m], t["cigars"][lo:lo + m]), (i, typ, v)
                    checked += 1; traced += int(want[0])
    assert checked == n * 6 and traced > 1500


def test_full_sw_traceback_golden(amd, fswtb_golden):
    """nvbio_full_sw_traceback (the linear-gap Smith-Waterman / edit-distance aligners through BatchedAlignmentTraceback) against the
    reference's own outputs (alignment_traceback<256,1024,64> over SmithWatermanAligner): scores, sources, sinks, CIGARs; unequal
    deletion / insertion costs; LOCAL walks that stop at SINK cells; min_score limits; scores and sinks computed here and handed over"""
    g = fswtb_golden
    max_p = int(np.diff(g["pat_off"]).max()); max_t = int(np.diff(g["txt_off"]).max())
    STRIDE = 128
    seen = traced = 0
    for typ in range(3):
        for si in range(len(g["schemes"])):
            sel = np.nonzero((g["typ"] == typ) & (g["scheme"] == si))[0].astype(np.uint32)
            if len(sel) == 0:
                continue
            batch = amd.AlignmentBatch(g["pats"], 8, g["pat_off"], g["txts"], 8, g["txt_off"][sel], g["txt_off"][sel + 1], read_id=sel)
            al = amd.make_smith_waterman_aligner(typ, amd.SimpleSmithWatermanScheme(*[int(v) for v in g["schemes"][si]]))
            ms = g["min_score"][sel].astype(np.int32)
            tb = amd.BatchedAlignmentTraceback(al)
            outs = [tb.enact(batch, max_p, max_t, min_scores=ms, cigar_stride=STRIDE)]
            # ... and with the scoring pass run by the caller (pattern blocking, as the traceback's own)
            s0, k0 = amd.BatchedAlignmentScore(al, text_blocking=False).enact(batch, max_p, max_t, min_scores=ms)
            outs.append(tb.enact(batch, max_p, max_t, min_scores=ms, cigar_stride=STRIDE, scores=s0, sinks=k0))
            for sc, src, snk, cig, ln in outs:
                sc, src, snk, ln = _i64(sc), amd.u32(src).astype(np.int64), amd.u32(snk).astype(np.int64), _i64(ln)
                cig = cig.cpu().numpy().view(np.uint16)
                for k, i in enumerate(sel):
                    want = g["out"][i]
                    assert sc[k] == want[1], (i, typ, si)
                    if want[0]:
                        lo, hi = int(g["cig_off"][i]), int(g["cig_off"][i + 1])
                        assert tuple(src[k]) == (want[2], want[3]) and tuple(snk[k]) == (want[4], want[5]), (i, typ, si)
                        assert ln[k] == hi - lo and np.array_equal(cig[k, :min(hi - lo, STRIDE)], g["cigars"][lo:lo + min(hi - lo, STRIDE)]), (i, typ, si)
                    else:
                        assert ln[k] == 0, (i, typ, si)
            seen += len(sel); traced += int(g["out"][sel, 0].sum())
    assert seen == len(g["typ"]) and traced > 1900


@pytest.mark.parametrize("typ", ["LOCAL", "SEMI_GLOBAL"])
def test_full_traceback_opposite_mate_shape(amd, orc, typ, tb_mode):
    """4-bit reads (reversed / complemented, qualities) in 2-bit genome windows of 150-500 symbols, chunked scratch,
    scores handed over from the scoring pass: equal to the oracle job by job"""
    import torch
    typ = getattr(oracle, typ)
    rng = np.random.default_rng(19)
    G = 200000
    text = rng.integers(0, 4, G, dtype=np.uint8)
    R, M = 900, 150
    starts = rng.integers(0, G - 600, R)
    off = rng.integers(0, 300, R)
    reads = []
    for j in range(R):
        r = text[starts[j] + off[j]:starts[j] + off[j] + M].copy()
        k = int(rng.integers(0, 4)); pos = rng.integers(0, M, k); r[pos] = (r[pos] + 1 + rng.integers(0, 3, k)) % 4
        if j % 4 == 0:
            c = int(rng.integers(3, M - 3)); gsz = int(rng.integers(1, 4))
            r = np.concatenate([r[:c], r[c + gsz:], rng.integers(0, 4, gsz, dtype=np.uint8)]) if j % 8 else \
                np.concatenate([r[:c], rng.integers(0, 4, gsz, dtype=np.uint8), r[c:M - gsz]])
        reads.append(r.astype(np.uint8))
    flags = rng.integers(0, 4, R).astype(np.uint8)
    stored = []
    for j, r in enumerate(reads):
        v = np.where(r < 4, 3 - r, r).astype(np.uint8) if flags[j] & 2 else r.copy()
        stored.append(v[::-1] if flags[j] & 1 else v)
    flat = np.concatenate(stored)
    quals = rng.integers(0, 64, len(flat), dtype=np.uint8)
    roffs = (np.arange(R + 1) * M).astype(np.uint32)
    wb = starts.astype(np.uint32); we = (starts + rng.integers(460, 501, R)).astype(np.uint32)
    we[::9] = wb[::9] + rng.integers(100, 200, len(we[::9])).astype(np.uint32)
    sv = (2, 2, 6, -8, -3, -8, -3) if typ == oracle.LOCAL else (0, 6, 6, -8, -3, -8, -3)
    ms = np.full(R, 60 if typ == oracle.LOCAL else -60, dtype=np.int32); ms[::3] = oracle.SCORE_MIN
    batch = amd.AlignmentBatch(orc.pack4(flat), 4, roffs, orc.pack2(text), 2, wb, we, quals=quals, flags=flags)
    al = amd.make_gotoh_aligner(typ, _scheme(amd, sv))
    op = amd.BatchedAlignmentTraceback(al)
    need = op.min_temp_storage(batch, M, 500)
    assert need == -(-R // 64) * 64 * 500 * 4 * (1 + 19)              # whole waves of 64 jobs own scratch
    temp = torch.empty(need // 3 + 64, dtype=torch.uint8, device="cuda:0")
    s0, k0 = amd.BatchedAlignmentScore(al, text_blocking=False).enact(batch, M, 500, min_scores=ms)
    for kw in (dict(), dict(temp=temp), dict(scores=s0, sinks=k0)):
        sc, src, snk, cig, ln = op.enact(batch, M, 500, min_scores=ms, cigar_stride=40, **kw)
        sc, src, snk, ln = sc.cpu().numpy(), amd.u32(src), amd.u32(snk), amd.u32(ln)
        cig = cig.cpu().numpy().view(np.uint16)
        for j in range(R):
            q = quals[roffs[j]:roffs[j + 1]]
            ok, s_, wsrc, wsnk, wc = orc.full_gotoh_traceback(typ, oracle.Scheme(*sv), reads[j], text[wb[j]:we[j]],
                                                             q[::-1] if flags[j] & 1 else q, int(ms[j]))
            assert sc[j] == s_ and tuple(snk[j]) == wsnk and tuple(src[j]) == wsrc and ln[j] == len(wc), (j, kw.keys())
            m = min(int(ln[j]), 40)                                 # elements beyond the stride are dropped, the count is not
            assert np.array_equal(cig[j, :m], wc[:m]), j
