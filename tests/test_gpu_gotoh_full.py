"""GPU parity: batched full-matrix Gotoh (text- and pattern-blocking, the short2 boundary column,
the stripe early exit) through the C-ABI vs the reference's golden vectors and the oracle."""
import numpy as np
import pytest

import oracle

pytestmark = pytest.mark.gpu


def _scheme(amd, v):
    return amd.GotohScheme(*[int(x) for x in v])


@pytest.mark.parametrize("blocking", [0, 1])
def test_full_golden(amd, dp_golden, blocking):
    g = dp_golden
    S = len(g["schemes"])
    n = len(g["pat_off"]) - 1
    lens_p = np.diff(g["pat_off"]); lens_t = np.diff(g["txt_off"])
    for si in range(S):
        for hq in (0, 1):
            cases = np.array([i for i in range(n) if i % S == si and int(g["has_quals"][i]) == hq], dtype=np.uint32)
            if len(cases) == 0:
                continue
            batch = amd.AlignmentBatch(g["pats"], 8, g["pat_off"], g["txts"], 8, g["txt_off"][cases],
                                       g["txt_off"][cases + 1], quals=g["quals"] if hq else None, read_id=cases)
            for typ in range(3):
                al = amd.make_gotoh_aligner(typ, _scheme(amd, g["schemes"][si]))
                for v in (0, 1):
                    ms = g["min_scores"][cases] if v else None
                    sc, sk = amd.BatchedAlignmentScore(al, text_blocking=bool(blocking)).enact(
                        batch, int(lens_p.max()), int(lens_t.max()), min_scores=ms)
                    want = g["full"][cases, blocking, typ, v]
                    assert np.array_equal(sc.cpu().numpy().astype(np.int64), want[:, 1]), (blocking, si, hq, typ, v)
                    assert np.array_equal(amd.u32(sk).astype(np.int64), want[:, 2:4]), (blocking, si, hq, typ, v)


@pytest.mark.parametrize("blocking", [0, 1])
def test_full_smith_waterman_and_edit_distance_golden(amd, orc, dp_golden, sw_golden, blocking):
    """the full-matrix SmithWatermanAligner and EditDistanceAligner (nvbio_full_sw_score): logical stripes of 16 -- LOCAL
    ties, the pattern-blocking early exit, none with text blocking -- on the reference's own outputs; then unequal
    deletion / insertion costs and tie-rich repetitive texts against the oracle's restatement of sw/sw_inl.h"""
    g, w = dp_golden, sw_golden
    n = len(g["pat_off"]) - 1
    lens_p = np.diff(g["pat_off"]); lens_t = np.diff(g["txt_off"])
    cases = np.arange(n, dtype=np.uint32)
    batch = amd.AlignmentBatch(g["pats"], 8, g["pat_off"], g["txts"], 8, g["txt_off"][cases], g["txt_off"][cases + 1], read_id=cases)
    ed_ms = np.maximum(g["min_scores"].astype(np.int64), -(lens_p.astype(np.int64) // 4) - 1).astype(np.int32)
    for typ in range(3):
        for v in (0, 1):
            for si, sw in enumerate(w["schemes"]):
                al = amd.make_smith_waterman_aligner(typ, amd.SimpleSmithWatermanScheme(*[int(x) for x in sw]))
                sc, sk = amd.BatchedAlignmentScore(al, text_blocking=bool(blocking)).enact(
                    batch, int(lens_p.max()), int(lens_t.max()), min_scores=g["min_scores"] if v else None)
                want = w["fsw"][:, si, blocking, typ, v]
                assert np.array_equal(sc.cpu().numpy().astype(np.int64), want[:, 1]), (blocking, si, typ, v)
                assert np.array_equal(amd.u32(sk).astype(np.int64), want[:, 2:4] & 0xFFFFFFFF), (blocking, si, typ, v)
            sc, sk = amd.BatchedAlignmentScore(amd.make_edit_distance_aligner(typ), text_blocking=bool(blocking)).enact(
                batch, int(lens_p.max()), int(lens_t.max()), min_scores=ed_ms if v else None)
            want = w["fed"][:, blocking, typ, v]
            assert np.array_equal(sc.cpu().numpy().astype(np.int64), want[:, 1]), (blocking, "ed", typ, v)
            assert np.array_equal(amd.u32(sk).astype(np.int64), want[:, 2:4] & 0xFFFFFFFF), (blocking, "ed", typ, v)
    rng = np.random.default_rng(23)
    G = 60000
    text = rng.integers(0, 4, G, dtype=np.uint8)
    text[10000:30000] = np.tile(rng.integers(0, 4, 7, dtype=np.uint8), 20000 // 7 + 1)[:20000]      # period 7: ties everywhere
    R = 1200
    lens = rng.integers(20, 140, R)
    roffs = np.zeros(R + 1, dtype=np.uint32); roffs[1:] = np.cumsum(lens)
    starts = rng.integers(100, G - 700, R)
    flat = np.concatenate([text[s + 40:s + 40 + l] for s, l in zip(starts, lens)]).copy()
    flat[rng.random(len(flat)) < 0.04] = rng.integers(0, 4)
    wb = starts.astype(np.uint32); we = (starts + rng.integers(150, 420, R)).astype(np.uint32)
    ms = rng.integers(-60, 120, R).astype(np.int32)
    batch = amd.AlignmentBatch(orc.pack4(flat), 4, roffs, orc.pack2(text), 2, wb, we)
    for sw in ((2, -3, -5, -2), (1, -2, -1, -4), (0, -1, -1, -1), (3, -2, -3, -3)):
        for typ in range(3):
            al = amd.make_smith_waterman_aligner(typ, amd.SimpleSmithWatermanScheme(*sw))
            sc, sk = amd.BatchedAlignmentScore(al, text_blocking=bool(blocking)).enact(batch, int(lens.max()), 420, min_scores=ms)
            sc, sk = sc.cpu().numpy(), amd.u32(sk)
            for k in range(R):
                ok, s, snk = orc.full_sw(typ, blocking, sw, flat[roffs[k]:roffs[k + 1]], text[wb[k]:we[k]], int(ms[k]))
                assert (int(sc[k]), int(sk[k, 0]), int(sk[k, 1])) == (s, snk[0], snk[1]), (blocking, sw, typ, k)


def test_sw_benchmark_shape(amd, orc):
    """many reads against ONE reference text (sw-benchmark/sw-benchmark.cu:152-197): LCG-random
    100 bp patterns (alignment_test.cu:879-881), global/semi-global/local Gotoh (2,-1,-2,-1), text blocking"""
    def lcg(n, seed):
        out = np.zeros(n, dtype=np.uint8)
        s = seed
        for i in range(n):
            s = (s * 1664525 + 1013904223) & 0xFFFFFFFF
            out[i] = s % 4
        return out, s
    R, M, N = 1500, 100, 1024
    pats, s = lcg(R * M, 0)
    text, _ = lcg(N, s)
    roffs = (np.arange(R + 1) * M).astype(np.uint32)
    wb = np.zeros(R, dtype=np.uint32); we = np.full(R, N, dtype=np.uint32)
    batch = amd.AlignmentBatch(orc.pack4(pats), 4, roffs, orc.pack2(text), 2, wb, we)
    sv = (2, 1, 1, -2, -1, -2, -1)
    toffs = (np.arange(R + 1) * N).astype(np.uint32)
    for typ in range(3):
        sc, sk = amd.BatchedAlignmentScore(amd.make_gotoh_aligner(typ, _scheme(amd, sv)), text_blocking=True).enact(batch, M, N)
        wsc, wsk = orc.full_gotoh_batch(typ, 1, oracle.Scheme(*sv), pats, roffs, np.tile(text, R), toffs)
        assert np.array_equal(sc.cpu().numpy(), wsc), typ
        assert np.array_equal(amd.u32(sk), wsk), typ
        # sw-benchmark stores int16 scores (sw-benchmark.cu:197): they must fit
        assert np.abs(wsc).max() < 32768


def test_opposite_mate_shape(amd, orc):
    """nvBowtie's opposite-mate scoring: full DP of a 150 bp read inside a <= 500 bp window with a
    finite min_score (early exit), reads reversed/complemented, qualities"""
    rng = np.random.default_rng(17)
    G = 200000
    text = rng.integers(0, 4, G, dtype=np.uint8)
    R, M = 3000, 150
    starts = rng.integers(0, G - 600, R)
    reads = np.stack([text[s + 100:s + 100 + M] for s in starts]).copy()
    reads[rng.random(reads.shape) < 0.03] = rng.integers(0, 4)
    flags = rng.integers(0, 4, R).astype(np.uint8)
    quals = rng.integers(0, 50, R * M, dtype=np.uint8)
    wb = starts.astype(np.uint32); we = (starts + rng.integers(150, 500, R)).astype(np.uint32)
    ms = rng.integers(-100, 280, R).astype(np.int32)
    roffs = (np.arange(R + 1) * M).astype(np.uint32)
    batch = amd.AlignmentBatch(orc.pack4(reads.reshape(-1)), 4, roffs, orc.pack2(text), 2, wb, we, quals=quals, flags=flags)
    sv = (2, 2, 6, -8, -3, -8, -3)
    for blocking in (0, 1):
        for typ in (oracle.LOCAL, oracle.SEMI_GLOBAL):
            sc, sk = amd.BatchedAlignmentScore(amd.make_gotoh_aligner(typ, _scheme(amd, sv)), text_blocking=bool(blocking)).enact(
                batch, M, 500, min_scores=ms)
            got_s, got_k = sc.cpu().numpy(), amd.u32(sk)
            for j in range(0, R, 7):
                p = reads[j][::-1] if flags[j] & 1 else reads[j]
                q = quals[j * M:(j + 1) * M][::-1] if flags[j] & 1 else quals[j * M:(j + 1) * M]
                if flags[j] & 2:
                    p = 3 - p
                ok, s, k = orc.full_gotoh(typ, blocking, oracle.Scheme(*sv), p, text[wb[j]:we[j]], q, int(ms[j]))
                assert got_s[j] == s and tuple(got_k[j]) == k, (blocking, typ, j)


@pytest.mark.parametrize("shortcut", [True, False])
def test_end_to_end_full_dp_with_ungapped_shortcut(amd, orc, shortcut, monkeypatch):
    """nvBowtie's end-to-end scheme (match 0) through the full-matrix kernel: jobs whose best diagonal beats every
    gapped alignment are settled without a DP, the rest go through it over a job list -- scores and sinks of
    every job equal the reference algorithm's, with and without min_score (stripe early exit), for reads with 0-3
    mismatches, indels, N's, reversed / complemented mates, windows up to and beyond the shortcut's 528 symbols,
    windows shorter than the read, both blockings"""
    if not shortcut:
        monkeypatch.setattr(amd, "DEFAULT_ALGO_FLAGS", amd.ALN_NO_UNGAPPED_SCORE)
    rng = np.random.default_rng(23)
    G = 300000
    text = rng.integers(0, 4, G, dtype=np.uint8)
    text[1000:1400] = np.tile(text[1000:1020], 20)                  # a tandem repeat: several diagonals tie
    R, M = 1400, 150
    lens = np.full(R, M); lens[::9] = rng.integers(40, 161, len(lens[::9]))
    roffs = np.zeros(R + 1, dtype=np.uint32); roffs[1:] = np.cumsum(lens)
    starts = rng.integers(0, G - 700, R); starts[:40] = rng.integers(1000, 1200, 40)
    wlen = rng.integers(150, 529, R); wlen[::11] = rng.integers(529, 640, len(wlen[::11])); wlen[::17] = 528; wlen[::13] = rng.integers(60, 150, len(wlen[::13]))
    off = (rng.random(R) * np.maximum(wlen - lens, 1)).astype(np.int64)
    reads = []
    for j in range(R):
        p = starts[j] + off[j]
        r = text[p:p + lens[j]].copy()
        k = int(rng.integers(0, 4))
        if k:
            pos = rng.integers(0, lens[j], k); r[pos] = (r[pos] + 1 + rng.integers(0, 3, k)) % 4
        if j % 7 == 0:
            c = int(rng.integers(5, lens[j] - 5)); g = int(rng.integers(1, 4))
            r = np.concatenate([r[:c], r[c + g:], rng.integers(0, 4, g, dtype=np.uint8)]) if j % 2 else \
                np.concatenate([r[:c], rng.integers(0, 4, g, dtype=np.uint8), r[c:lens[j] - g]])
        if j % 5 == 3:
            # the only difference is a 1-4 bp indel a few bases from an end: the best diagonal has 2-3 mismatches but
            # a single-gap alignment scores higher -- the shortcut's second chance must hand these to the DP
            r = text[p:p + lens[j]].copy(); src = text[p:p + lens[j] + 8]
            g = int(rng.integers(1, 5)); e = int(rng.integers(1, 7)); at = e if rng.random() < 0.5 else lens[j] - e
            r = np.concatenate([src[:at], src[at + g:]])[:lens[j]] if rng.random() < 0.5 else \
                np.concatenate([src[:at], rng.integers(0, 4, g, dtype=np.uint8), src[at:]])[:lens[j]]
        if j % 29 == 0:
            r[int(rng.integers(0, lens[j]))] = 4
        reads.append(r.astype(np.uint8))
    flags = rng.integers(0, 4, R).astype(np.uint8)
    # store each read so that the flagged view (reverse / complement) is the mutated locus
    stored = []
    for j, r in enumerate(reads):
        v = r.copy()
        if flags[j] & 2:
            v = np.where(v < 4, 3 - v, v).astype(np.uint8)
        if flags[j] & 1:
            v = v[::-1]
        stored.append(v)
    flat = np.concatenate(stored)
    quals = rng.integers(0, 64, len(flat), dtype=np.uint8)
    wb = starts.astype(np.uint32); we = np.minimum(starts + wlen, G).astype(np.uint32)
    ms = rng.integers(-60, 1, R).astype(np.int32); ms[::5] = oracle.SCORE_MIN
    sv = (0, 6, 6, -8, -3, -8, -3)
    settled = 0
    for use_q in (False, True):
        for min_scores in (None, ms):
            batch = amd.AlignmentBatch(orc.pack4(flat), 4, roffs, orc.pack2(text), 2, wb, we, quals=quals if use_q else None, flags=flags)
            for blocking in (1, 0):
                sc, sk = amd.BatchedAlignmentScore(amd.make_gotoh_aligner(oracle.SEMI_GLOBAL, _scheme(amd, sv)), text_blocking=bool(blocking)).enact(
                    batch, 161, 640, min_scores=min_scores)
                got_s, got_k = sc.cpu().numpy(), amd.u32(sk)
                for j in range(R):
                    q = quals[roffs[j]:roffs[j + 1]]
                    if flags[j] & 1:
                        q = q[::-1]
                    ok, s_, k_ = orc.full_gotoh(oracle.SEMI_GLOBAL, blocking, oracle.Scheme(*sv), reads[j], text[wb[j]:we[j]],
                                               q if use_q else None, int(min_scores[j]) if min_scores is not None else oracle.SCORE_MIN)
                    assert got_s[j] == s_ and tuple(got_k[j]) == k_, (use_q, min_scores is not None, blocking, j, lens[j], wlen[j])
                    settled += int(s_ > -8)
    assert settled > 2000                                           # plenty of jobs the shortcut can settle


@pytest.mark.parametrize("narrow", [True, False])
def test_narrow_route_and_the_other_locus(amd, orc, narrow, monkeypatch):
    """end-to-end full-matrix scoring, pattern blocking: the jobs the shortcut cannot settle are scored in a band of 31 diagonals around
    their best diagonal and keep that score only where no other diagonal of the window holds a long exact run (narrow_check_kernel).
    Here the window holds the read twice: locus A with an indel of 1-3 symbols and 0-2 substitutions, locus B -- 40-250 symbols away,
    sometimes only 5-20 -- as a copy with 2-5 substitutions, so that the best DIAGONAL is B's while the best ALIGNMENT is either's;
    plus reads with 3-6 substitutions and no second locus, and unrelated reads with a per-job min_score.  Scores and sinks equal the
    reference algorithm's with and without the route (NVBIO_ALN_NO_NARROW_SCORE)."""
    monkeypatch.setattr(amd, "DEFAULT_ALGO_FLAGS", amd.ALN_FORCE_PACKED_DP | (0 if narrow else amd.ALN_NO_NARROW_SCORE))
    rng = np.random.default_rng(211)
    R, W = 1500, 500
    G = R * 560
    text = rng.integers(0, 4, G, dtype=np.uint8)
    lens = np.full(R, 150); lens[3::11] = rng.integers(60, 150, len(lens[3::11]))
    roffs = np.zeros(R + 1, dtype=np.uint32); roffs[1:] = np.cumsum(lens)
    wb = (np.arange(R) * 560 + 30).astype(np.uint32); we = (wb + W).astype(np.uint32)
    we[5::13] = wb[5::13] + rng.integers(200, 500, len(we[5::13]))
    reads = []
    for j in range(R):
        M = int(lens[j]); N = int(we[j] - wb[j])
        a = int(wb[j]) + int(rng.integers(0, N - M - 4))
        src = text[a:a + M + 4]
        kind = j % 4
        if kind == 3:
            r = rng.integers(0, 4, M).astype(np.uint8)             # unrelated
        else:
            r = src[:M].copy()
            if kind in (0, 1):                                      # an indel at locus A
                cpos = int(rng.integers(5, M - 5)); g = int(rng.integers(1, 4))
                r = np.concatenate([src[:cpos], src[cpos + g:]])[:M] if rng.random() < 0.5 else \
                    np.concatenate([src[:cpos], rng.integers(0, 4, g, dtype=np.uint8), src[cpos:]])[:M]
                k = int(rng.integers(0, 3))
            else:
                k = int(rng.integers(3, 7))                         # substitutions only, more than the shortcut takes
            pos = rng.choice(M, k, replace=False); r[pos] = (r[pos] + 1 + rng.integers(0, 3, k)) % 4
            if kind == 0:                                           # ... and a second locus B inside the window
                sep = int(rng.integers(5, 21)) if rng.random() < 0.3 else int(rng.integers(40, 251))
                bpos = a + sep if a + sep + M <= int(we[j]) else a - sep
                if bpos >= int(wb[j]) and bpos + M <= int(we[j]):
                    c = r.copy(); kk = int(rng.integers(2, 6)); pos = rng.choice(M, kk, replace=False); c[pos] = (c[pos] + 1 + rng.integers(0, 3, kk)) % 4
                    text[bpos:bpos + M] = c
                    r2 = text[a:a + M + 4]                          # (B may have overwritten part of A: the oracle decides what is best)
        reads.append(r.astype(np.uint8))
    flat = np.concatenate(reads)
    sv = (0, 6, 6, -8, -3, -8, -3)
    ms = rng.integers(-70, -10, R).astype(np.int32); ms[::3] = oracle.SCORE_MIN
    batch = amd.AlignmentBatch(orc.pack4(flat), 4, roffs, orc.pack2(text), 2, wb, we)
    gapped = deep = 0
    for min_scores in (None, ms):
        sc, sk = amd.BatchedAlignmentScore(amd.make_gotoh_aligner(oracle.SEMI_GLOBAL, _scheme(amd, sv)), text_blocking=False).enact(batch, 150, W, min_scores=min_scores)
        got_s, got_k = sc.cpu().numpy(), amd.u32(sk)
        for j in range(R):
            ok, s_, k_ = orc.full_gotoh(oracle.SEMI_GLOBAL, 0, oracle.Scheme(*sv), reads[j], text[wb[j]:we[j]], None,
                                       int(min_scores[j]) if min_scores is not None else oracle.SCORE_MIN)
            assert got_s[j] == s_ and tuple(got_k[j]) == k_, (narrow, min_scores is not None, j, j % 4, lens[j])
            gapped += int(-40 < s_ < -7 and s_ % 6 != 0); deep += int(s_ <= -18)
    assert gapped > 800 and deep > 400


@pytest.mark.parametrize("narrow", [True, False])
def test_full_e2e_scoring_on_low_complexity_text(amd, orc, narrow, monkeypatch):
    """end-to-end full-matrix scoring where equal scores and long exact runs are everywhere: tandem repeats of period 1-7 with a few
    mutations, two-letter stretches, reads that are themselves periodic (so that many diagonals and many gapped alignments tie and the
    sink is decided by "the last of equal scores"), with 0-5 substitutions and an occasional indel; both blockings; with and without
    the narrow route.  Scores AND sinks equal the reference algorithm's."""
    monkeypatch.setattr(amd, "DEFAULT_ALGO_FLAGS", amd.ALN_FORCE_PACKED_DP | (0 if narrow else amd.ALN_NO_NARROW_SCORE))
    rng = np.random.default_rng(313)
    R, W = 1200, 420
    G = R * 480
    text = rng.integers(0, 4, G, dtype=np.uint8)
    lens = np.full(R, 100); lens[2::9] = rng.integers(40, 151, len(lens[2::9]))
    roffs = np.zeros(R + 1, dtype=np.uint32); roffs[1:] = np.cumsum(lens)
    wb = (np.arange(R) * 480 + 20).astype(np.uint32); we = (wb + W).astype(np.uint32)
    reads = []
    for j in range(R):
        M = int(lens[j])
        lo, hi = int(wb[j]), int(we[j])
        kind = j % 3
        if kind == 0:                                               # a tandem repeat fills most of the window
            unit = rng.integers(0, 4, int(rng.integers(1, 8))).astype(np.uint8)
            a0 = lo + int(rng.integers(0, 60)); L = int(rng.integers(M + 20, hi - a0))
            rep = np.resize(unit, L).copy()
            mut = rng.random(L) < 0.02; rep[mut] = rng.integers(0, 4, int(mut.sum()))
            text[a0:a0 + L] = rep
        elif kind == 1:                                             # a two-letter stretch
            a0 = lo + int(rng.integers(0, 100)); L = int(rng.integers(M, hi - a0))
            text[a0:a0 + L] = rng.integers(0, 2, L) * int(rng.integers(1, 4))
        a = lo + int(rng.integers(0, W - M - 4))
        src = text[a:a + M + 4]
        r = src[:M].copy()
        if rng.random() < 0.3:
            cpos = int(rng.integers(3, M - 3)); g = int(rng.integers(1, 4))
            r = np.concatenate([src[:cpos], src[cpos + g:]])[:M] if rng.random() < 0.5 else np.concatenate([src[:cpos], src[cpos - g:cpos], src[cpos:]])[:M]
        k = int(rng.integers(0, 6))
        if k:
            pos = rng.choice(M, k, replace=False); r[pos] = (r[pos] + 1 + rng.integers(0, 3, k)) % 4
        reads.append(r.astype(np.uint8))
    flat = np.concatenate(reads)
    sv = (0, 6, 6, -8, -3, -8, -3)
    batch = amd.AlignmentBatch(orc.pack4(flat), 4, roffs, orc.pack2(text), 2, wb, we)
    unsettled = 0
    for blocking in (0, 1):
        sc, sk = amd.BatchedAlignmentScore(amd.make_gotoh_aligner(oracle.SEMI_GLOBAL, _scheme(amd, sv)), text_blocking=bool(blocking)).enact(batch, 150, W)
        got_s, got_k = sc.cpu().numpy(), amd.u32(sk)
        for j in range(R):
            ok, s_, k_ = orc.full_gotoh(oracle.SEMI_GLOBAL, blocking, oracle.Scheme(*sv), reads[j], text[wb[j]:we[j]], None, oracle.SCORE_MIN)
            assert got_s[j] == s_ and tuple(got_k[j]) == k_, (narrow, blocking, j, j % 3, lens[j])
            unsettled += int(s_ <= -8)
    assert unsettled > 600


def test_shortcut_sees_alignments_hanging_over_the_window_ends(amd, orc):
    """end-to-end full-matrix scoring: the window holds a copy of the read with two substitutions (best diagonal: -12), while at one of
    its ENDS all but the read's first / last symbol match -- one inserted symbol, -8, on a diagonal that is not wholly inside the
    window.  The shortcut's second chance must hand these to the DP (it used to look at whole diagonals only); both blockings, with the
    hang at either end, patterns of 20-150 symbols"""
    rng = np.random.default_rng(97)
    R, W = 800, 320
    G = R * 400
    text = rng.integers(0, 4, G, dtype=np.uint8)
    lens = rng.integers(20, 151, R)
    roffs = np.zeros(R + 1, dtype=np.uint32); roffs[1:] = np.cumsum(lens)
    wb = (np.arange(R) * 400 + 20).astype(np.uint32); we = (wb + W).astype(np.uint32)
    reads = []
    for j in range(R):
        M = int(lens[j])
        if j % 2:                                                   # the read's last symbol hangs over the window's end
            r = np.concatenate([text[we[j] - (M - 1):we[j]], [(text[we[j]] + 1) % 4]]).astype(np.uint8)
        else:                                                       # ... its first symbol over the window's beginning
            r = np.concatenate([[(text[wb[j] - 1] + 1) % 4], text[wb[j]:wb[j] + M - 1]]).astype(np.uint8)
        if j % 7:                                                   # a copy with two substitutions near the window's other end
            c = r.copy(); pos = rng.choice(M, 2, replace=False); c[pos] = (c[pos] + 1) % 4
            at = int(wb[j]) + 5 if j % 2 else int(we[j]) - M - 5
            text[at:at + M] = c
        reads.append(r)
    flat = np.concatenate(reads)
    sv = (0, 6, 6, -8, -3, -8, -3)
    batch = amd.AlignmentBatch(orc.pack4(flat), 4, roffs, orc.pack2(text), 2, wb, we)
    gapped = 0
    for blocking in (0, 1):
        sc, sk = amd.BatchedAlignmentScore(amd.make_gotoh_aligner(oracle.SEMI_GLOBAL, _scheme(amd, sv)), text_blocking=bool(blocking)).enact(batch, 150, W)
        got_s, got_k = sc.cpu().numpy(), amd.u32(sk)
        for j in range(R):
            ok, s_, k_ = orc.full_gotoh(oracle.SEMI_GLOBAL, blocking, oracle.Scheme(*sv), reads[j], text[wb[j]:we[j]], None, oracle.SCORE_MIN)
            assert got_s[j] == s_ and tuple(got_k[j]) == k_, (blocking, j, lens[j])
            gapped += int(s_ == -8)
    assert gapped > 1200                                            # the hanging alignment wins nearly everywhere


@pytest.mark.parametrize("typ,M", [("LOCAL", 150), ("SEMI_GLOBAL", 150), ("GLOBAL", 150), ("SEMI_GLOBAL", 140), ("SEMI_GLOBAL", 9), ("SEMI_GLOBAL", 24)])
def test_packed_pattern_blocking_kernel(amd, orc, typ, M, monkeypatch):
    """pattern blocking on a batch of one dominant shape (150 x 400, the opposite-mate case): those jobs run two per
    lane in 16-bit registers, odd-shaped ones through the int32 kernel; odd job counts, reversed / complemented
    reads, qualities, N's, per-job min_score (early exit of one job of a pair only), with and without the
    end-to-end shortcut in front -- every score and sink equals the reference algorithm's"""
    typ = getattr(oracle, typ)
    monkeypatch.setattr(amd, "DEFAULT_ALGO_FLAGS", amd.ALN_FORCE_PACKED_DP)    # the library keeps small batches on the int32 kernel
    rng = np.random.default_rng(41)
    G = 200000
    text = rng.integers(0, 4, G, dtype=np.uint8)
    # (M = 140, 9, 24: the end-to-end kernel sweeps 16 columns at a time and the reference tests its early exit every 8: a real 8-column
    # boundary inside the last stripe, a pattern of two blocks of 8 in one stripe, a last stripe that is half empty)
    R, W = 701, 400
    lens = np.full(R, M); lens[5::50] = rng.integers(min(60, M - 1), M, len(lens[5::50]))
    roffs = np.zeros(R + 1, dtype=np.uint32); roffs[1:] = np.cumsum(lens)
    starts = rng.integers(0, G - 500, R)
    wlen = np.full(R, W); wlen[7::40] = rng.integers(150, 400, len(wlen[7::40]))
    off = rng.integers(0, 200, R)
    reads = []
    for j in range(R):
        r = text[starts[j] + off[j]:starts[j] + off[j] + lens[j]].copy()
        k = int(rng.integers(0, 6))
        if k:
            pos = rng.integers(0, lens[j], k); r[pos] = (r[pos] + 1 + rng.integers(0, 3, k)) % 4
        if j % 5 == 0 and lens[j] > 12:
            c = int(rng.integers(5, lens[j] - 5)); g = int(rng.integers(1, 5))
            r = np.concatenate([r[:c], r[c + g:], rng.integers(0, 4, g, dtype=np.uint8)]) if j % 2 else \
                np.concatenate([r[:c], rng.integers(0, 4, g, dtype=np.uint8), r[c:lens[j] - g]])
        if j % 31 == 0:
            r[int(rng.integers(0, lens[j]))] = 4
        if j % 23 == 0:
            r = rng.integers(0, 4, lens[j]).astype(np.uint8)        # unrelated: exercises the early exit
        reads.append(r.astype(np.uint8))
    flags = rng.integers(0, 4, R).astype(np.uint8)
    stored = []
    for j, r in enumerate(reads):
        v = r.copy()
        if flags[j] & 2:
            v = np.where(v < 4, 3 - v, v).astype(np.uint8)
        if flags[j] & 1:
            v = v[::-1]
        stored.append(v)
    flat = np.concatenate(stored)
    quals = rng.integers(0, 64, len(flat), dtype=np.uint8)
    wb = starts.astype(np.uint32); we = (starts + wlen).astype(np.uint32)
    for sv in ((2, 2, 6, -8, -3, -8, -3), (0, 6, 6, -8, -3, -8, -3)):
        if typ == oracle.LOCAL and sv[0] == 0:
            continue
        lo = -100 if sv[0] == 0 else 100
        ms = rng.integers(lo, lo + 150, R).astype(np.int32); ms[::4] = oracle.SCORE_MIN
        for use_q in (True, False):
            batch = amd.AlignmentBatch(orc.pack4(flat), 4, roffs, orc.pack2(text), 2, wb, we, quals=quals if use_q else None, flags=flags)
            for min_scores in (ms, None):
                sc, sk = amd.BatchedAlignmentScore(amd.make_gotoh_aligner(typ, _scheme(amd, sv)), text_blocking=False).enact(
                    batch, M, W, min_scores=min_scores)
                got_s, got_k = sc.cpu().numpy(), amd.u32(sk)
                for j in range(R):
                    q = quals[roffs[j]:roffs[j + 1]]
                    if flags[j] & 1:
                        q = q[::-1]
                    ok, s_, k_ = orc.full_gotoh(typ, 0, oracle.Scheme(*sv), reads[j], text[wb[j]:we[j]], q if use_q else None,
                                               int(min_scores[j]) if min_scores is not None else oracle.SCORE_MIN)
                    assert got_s[j] == s_ and tuple(got_k[j]) == k_, (sv, use_q, min_scores is not None, j, lens[j], wlen[j])


@pytest.mark.parametrize("max_m", [8, 32, 64, 100, 128, 152, 200, 256])
def test_cooperative_kernel_equals_the_oracle(amd, orc, max_m):
    """full_gotoh_coop_kernel (several lanes per job, the boundary column in registers): GLOBAL and SEMI_GLOBAL without early exit, every
    pattern-length bucket of its launcher with ragged patterns (1 .. max_m) and windows (0 .. 300 symbols), reversed / complemented reads, N
    symbols, base qualities under a quality ramp, border gap terms that differ between pattern and text, both blockings, batches that end
    inside a wave -- equal to the oracle job by job and to the one-lane-per-job kernel (NVBIO_ALN_NO_COOPERATIVE_DP)"""
    rng = np.random.default_rng(1000 + max_m)
    G = 50000
    text = rng.integers(0, 4, G, dtype=np.uint8)
    for R in (1, 3, 517):
        lens = rng.integers(1, max_m + 1, R); lens[0] = max_m
        roffs = np.zeros(R + 1, dtype=np.uint32); roffs[1:] = np.cumsum(lens)
        wb = rng.integers(0, G - 400, R).astype(np.uint32)
        wl = rng.integers(1, 300, R); wl[R // 2] = 0 if R > 1 else wl[R // 2]
        we = (wb + wl).astype(np.uint32)
        reads = []
        for k in range(R):
            src = text[wb[k] + 5:wb[k] + 5 + lens[k]].copy() if rng.random() < 0.7 else rng.integers(0, 4, lens[k], dtype=np.uint8)
            if len(src) < lens[k]:
                src = rng.integers(0, 4, lens[k], dtype=np.uint8)
            mut = rng.random(lens[k]) < 0.05; src[mut] = rng.integers(0, 4, int(mut.sum()))
            if rng.random() < 0.2:
                src[int(rng.integers(0, lens[k]))] = 4
            reads.append(src.astype(np.uint8))
        flat = np.concatenate(reads)
        flags = rng.integers(0, 4, R).astype(np.uint8)
        quals = rng.integers(0, 50, len(flat), dtype=np.uint8)
        toffs = np.zeros(R + 1, dtype=np.uint32); toffs[1:] = np.cumsum(we - wb)
        txts = np.concatenate([text[wb[k]:we[k]] for k in range(R)] + [np.zeros(0, dtype=np.uint8)])
        # the oracle takes patterns as they are aligned: apply the read flags and the matching quality order
        pats_o, quals_o = [], []
        for k in range(R):
            p_, q_ = reads[k], quals[roffs[k]:roffs[k + 1]]
            if flags[k] & 1:
                p_, q_ = p_[::-1], q_[::-1]
            if flags[k] & 2:
                p_ = np.where(p_ < 4, 3 - p_, p_)
            pats_o.append(p_.astype(np.uint8)); quals_o.append(q_)
        pats_o = np.concatenate(pats_o); quals_o = np.concatenate(quals_o)
        for sv, use_q in (((2, 1, 1, -2, -1, -2, -1), False), ((0, 2, 6, -8, -3, -5, -2), True), ((1, 3, 3, -4, -2, -7, -1), False)):
            for blocking in (0, 1):
                for typ in (oracle.GLOBAL, oracle.SEMI_GLOBAL):
                    if typ == oracle.SEMI_GLOBAL and sv[0] == 0:
                        continue                                            # (match = 0 end-to-end: the shortcut's route, tested elsewhere)
                    wsc, wsk = orc.full_gotoh_batch(typ, blocking, oracle.Scheme(*sv), pats_o, roffs, txts, toffs, quals=quals_o if use_q else None)
                    for algo in (None, amd.ALN_NO_COOPERATIVE_DP):
                        batch = amd.AlignmentBatch(orc.pack4(flat), 4, roffs, orc.pack2(text), 2, wb, we, quals=quals if use_q else None, flags=flags,
                                                   algo_flags=algo)
                        sc, sk = amd.BatchedAlignmentScore(amd.make_gotoh_aligner(typ, _scheme(amd, sv)), text_blocking=bool(blocking)).enact(batch, max_m, 300)
                        bad = np.nonzero((sc.cpu().numpy() != wsc) | (amd.u32(sk) != wsk).any(axis=1))[0]
                        assert len(bad) == 0, (max_m, R, sv, blocking, typ, algo, bad[:5], sc.cpu().numpy()[bad[:5]], wsc[bad[:5]], amd.u32(sk)[bad[:5]], wsk[bad[:5]],
                                               lens[bad[:5]], (we - wb)[bad[:5]])
