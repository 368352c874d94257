"""The oracle (oracle/nvbio_oracle.c) against the golden vectors the REFERENCE produced
(tests/golden/*.npz, generator tests/golden/make_golden.py).  CPU only."""
import numpy as np

import oracle


def _mask_stale(bwt_occ, n):
    """gen_bwt_from_sa leaves a stale (n+1)-th symbol in the padding (nvbio/fmindex/bwt.h:41-53)"""
    b = bwt_occ.copy()
    w = (n >> 6) * 8 + ((n & 63) >> 4)
    if w < len(b):
        b[w] &= ~np.uint32(3 << (30 - 2 * (n & 15)))
    return b


def test_packed_stream_roundtrip(orc):
    # nvbio-test/packedstream_test.cpp: pack/unpack round trips, big-endian layout
    rng = np.random.default_rng(1)
    s = rng.integers(0, 4, 1003, dtype=np.uint8)
    w = orc.pack2(s)
    assert all(orc.get2(w, i) == s[i] for i in range(len(s)))
    assert (int(w[0]) >> 30) == s[0] and ((int(w[0]) >> 28) & 3) == s[1]
    s4 = rng.integers(0, 5, 777, dtype=np.uint8)
    w4 = orc.pack4(s4)
    assert all(orc.get4(w4, i) == s4[i] for i in range(len(s4)))
    assert (int(w4[0]) >> 28) == s4[0]


def test_sais_sanity_banana(orc):
    # nvbio-test/packedstream_test.cpp:143-172: SA("banana") == {5,3,1,0,4,2}; over a 2-bit
    # alphabet the same shape is a=0, b=1, n=2
    text = np.array([1, 0, 2, 0, 2, 0], dtype=np.uint8)
    sa = orc.suffix_sort(text)
    assert list(sa) == [6, 5, 3, 1, 0, 4, 2]


def test_suffix_sort_matches_reference_sa(orc, fm_golden):
    sa = orc.suffix_sort(fm_golden["text"])
    assert np.array_equal(sa[1:], fm_golden["sa"][1:])


def test_index_build_matches_reference(orc, fm_golden):
    g = fm_golden
    idx = orc.build_index(g["text"])
    assert idx.primary == int(g["primary"])
    assert np.array_equal(idx.L2, g["L2"])
    assert np.array_equal(idx.bwt_occ, _mask_stale(g["bwt_occ"], len(g["text"])))
    assert np.array_equal(idx.ssa, g["ssa"])
    assert idx.ssa[0] == 0xFFFFFFFF


def _golden_index(g):
    return oracle.HostIndex(len(g["text"]), int(g["primary"]), g["L2"], g["bwt_occ"], g["ssa"])


def test_rank_every_row(orc, fm_golden):
    # the shape of nvbio-test/rank_test.cu:46-79: every i, every c, rank and rank4
    g = fm_golden
    idx = _golden_index(g)
    n = idx.n
    for k in range(-1, n + 1):
        for c in range(4):
            assert orc.rank(idx, k, c) == g["ranks"][k + 1, c], (k, c)
        if k >= 0:
            assert np.array_equal(orc.rank4(idx, k), g["ranks4"][k]), k
    for (l, r, c), want in zip(g["rank2_args"], g["rank2"]):
        assert np.array_equal(orc.rank2(idx, int(l), int(r), int(c)), want), (l, r, c)


def test_rank_is_a_running_count(orc, fm_golden):
    # independent of the reference outputs: rank(k,c) = #c in the first k+1 BWT-matrix rows
    g = fm_golden
    idx = _golden_index(g)
    text, sa = g["text"], g["sa"]
    L = [-1 if sa[i] == 0 else int(text[sa[i] - 1]) for i in range(1, idx.n + 1)]
    L = [-1 if i == 0 else x for i, x in enumerate([0] + L)]  # row 0 = '$' suffix: preceded by text[n-1]
    L[0] = int(text[-1])
    counts = [0, 0, 0, 0]
    for k in range(idx.n + 1):
        if L[k] >= 0:
            counts[L[k]] += 1
        for c in range(4):
            assert orc.rank(idx, k, c) == counts[c], (k, c)


def test_match_ranges(orc, fm_golden):
    g = fm_golden
    idx = _golden_index(g)
    assert np.array_equal(orc.match_batch(idx, g["q_syms"], g["q_offs"], False), g["ranges_bwd"])
    assert np.array_equal(orc.match_batch(idx, g["q_syms"], g["q_offs"], True), g["ranges_fwd"])


def test_match_against_naive_count(orc, fm_golden):
    # match() == number of occurrences found by a naive scan (fmindex_test.cu:603-657 self-check)
    g = fm_golden
    idx = _golden_index(g)
    text = g["text"]
    ranges = orc.match_batch(idx, g["q_syms"], g["q_offs"], False)
    tb = text.tobytes()
    for q in range(0, 200):
        p = g["q_syms"][g["q_offs"][q]:g["q_offs"][q + 1]]
        if (p > 3).any():
            continue
        pb, cnt, st = p.tobytes(), 0, 0
        while True:
            st = tb.find(pb, st)
            if st < 0:
                break
            cnt += 1
            st += 1
        x, y = int(ranges[q, 0]), int(ranges[q, 1])
        assert (y + 1 - x if y >= x else 0) == cnt, q


def test_locate(orc, fm_golden):
    g = fm_golden
    idx = _golden_index(g)
    assert np.array_equal(orc.locate_batch(idx, g["rows"]), g["pos"])
    assert np.array_equal(orc.locate_ssa_batch(idx, g["rows"]), g["jt"])
    # locate(i) == SA[i] for every row but 0 (SSA test of fmindex_test.cu:575-585)
    rows = g["rows"][g["rows"] > 0]
    assert np.array_equal(orc.locate_batch(idx, rows), g["sa"][rows])


def test_filter_rank_and_locate(orc, fm_golden):
    g = fm_golden
    idx = _golden_index(g)
    total, ranges, slots = orc.filter_rank(idx, g["q_syms"], g["q_offs"])
    assert np.array_equal(ranges, g["ranges_bwd"])
    sizes = np.where(ranges[:, 1] >= ranges[:, 0], ranges[:, 1].astype(np.int64) + 1 - ranges[:, 0], 0)
    assert np.array_equal(slots, np.cumsum(sizes).astype(np.uint64))
    assert total == int(sizes.sum())
    hits = orc.filter_locate(idx, ranges, slots, 0, min(total, 2000))
    # every hit is an occurrence of its query
    text = g["text"]
    for pos, q in hits[:500]:
        p = g["q_syms"][g["q_offs"][q]:g["q_offs"][q + 1]]
        assert np.array_equal(text[pos:pos + len(p)], p)


def _case(g, i):
    p = g["pats"][g["pat_off"][i]:g["pat_off"][i + 1]]
    t = g["txts"][g["txt_off"][i]:g["txt_off"][i + 1]]
    q = g["quals"][g["pat_off"][i]:g["pat_off"][i + 1]] if g["has_quals"][i] else None
    return p, t, q


def test_banded_gotoh_golden(orc, dp_golden):
    g = dp_golden
    S = len(g["schemes"])
    n = len(g["pat_off"]) - 1
    for i in range(n):
        sc = oracle.Scheme(*[int(v) for v in g["schemes"][i % S]])
        p, t, q = _case(g, i)
        for bi, b in enumerate(g["bands"]):
            for typ in range(3):
                ok, s, sk = orc.banded_gotoh(int(b), typ, sc, p, t, q)
                want = g["banded"][i, bi, typ]
                assert (ok, s, sk[0], sk[1]) == tuple(int(v) for v in want), (i, b, typ)


def _tb_scheme(g, i):
    S = len(g["schemes"])
    v = g["known_schemes"][i] if i < int(g["n_known"]) else g["schemes"][i % S]
    return oracle.Scheme(*[int(x) for x in v])


def test_banded_traceback_golden(orc, dp_golden, tb_golden):
    """score, source, sink and run-length CIGAR of the reference's banded_alignment_traceback"""
    g, t = dp_golden, tb_golden
    n = len(g["pat_off"]) - 1
    u = lambda v: int(np.uint32(np.int64(v) & 0xFFFFFFFF))
    checked = 0
    for i in range(n):
        sc = _tb_scheme(g, i)
        p, x, q = _case(g, i)
        for bi, b in enumerate(t["bands"]):
            for typ in range(3):
                want = t["aln"][i, bi, typ]
                if want[0] < 0:
                    continue
                ok, s, src, snk, cig, ops = orc.banded_gotoh_traceback(int(b), typ, sc, p, x, q)
                assert (ok, s) == (int(want[0]), int(want[1])), (i, b, typ)
                assert src == (u(want[2]), u(want[3])) and snk == (u(want[4]), u(want[5])), (i, b, typ)
                lo, hi = t["cig_off"][i, bi, typ]
                assert np.array_equal(cig, t["cigars"][lo:hi]), (i, b, typ, oracle.cigar_string(cig))
                if ok:      # the CIGAR spans the pattern and the source/sink text interval
                    ln = cig >> 2; ty = cig & 3
                    assert int(ln[(ty == 0) | (ty == 1) | (ty == 3)].sum()) == len(p)
                    assert int(ln[(ty == 0) | (ty == 2)].sum()) == snk[0] - src[0]
                checked += 1
    assert checked > 4000
    # the reference's functional tests (alignment_test.cu:743-746,779-785)
    p, x, _ = _case(g, 0)
    r = orc.banded_gotoh_traceback(7, oracle.SEMI_GLOBAL, oracle.Scheme.simple(2, -1, -1, -1), p, x)
    assert oracle.cigar_string(r[4]) == "4M1D3M"
    p, x, _ = _case(g, 1)
    r = orc.banded_gotoh_traceback(31, oracle.SEMI_GLOBAL, oracle.Scheme.simple(0, -5, -8, -3), p, x)
    assert oracle.cigar_string(r[4]) == "147M2D3M" and r[1] == -11 and r[2] == (13, 0) and r[3] == (165, 150)


def test_banded_edit_distance_golden(orc, dp_golden, ed_golden):
    """the reference's banded edit-distance aligner (fmmap's, nvBowtie --scoring ed) == banded Gotoh with
    (match 0, mismatch -1, open = extension = -1): equal open and extension costs make the recurrences coincide"""
    g, e = dp_golden, ed_golden
    sc = oracle.Scheme(*oracle.ED_SCHEME)
    checked = 0
    for i in range(len(g["pat_off"]) - 1):
        p, t, _ = _case(g, i)
        for bi, b in enumerate(e["bands"]):
            for typ in range(3):
                want = e["ed"][i, bi, typ]
                if want[0] < 0:
                    continue
                ok, s, sk = orc.banded_gotoh(int(b), typ, sc, p, t)
                assert (ok, s, np.int64(np.int32(np.uint32(sk[0]))), np.int64(np.int32(np.uint32(sk[1])))) == tuple(int(v) for v in want), (i, b, typ)
                checked += 1
    assert checked > 4000
    # the reference's functional tests (alignment_test.cu:643-705): edit distances 0, -2, -2, 0, -2, -2
    def enc(x):
        return np.array(["ACGT".index(c) for c in x], dtype=np.uint8)
    for pat, txt, want in (("GGGTGCTCAA", "AAAAGGGTGCTCAA", 0), ("GGGTAAGCTC", "AAAAGGGTGCTCAA", -2), ("AAGGGTGCTC", "AAAAGGGTGCAATC", -2),
                           ("AAAAGGGTGC", "AAAAGGGTGCTCAA", 0), ("AAAAGGGTG", "AAAAGGAAGTGCTC", -2), ("CACCGGGT", "AACAGGGTGCTC", -2)):
        assert orc.banded_gotoh(7, oracle.SEMI_GLOBAL, sc, enc(pat), enc(txt))[1] == want


def test_smith_waterman_family_golden(orc, dp_golden, sw_golden):
    """the reference's linear-gap SmithWatermanAligner (banded and full matrix, both blockings, with and without a minimum
    score) and its full-matrix EditDistanceAligner, on the reference's own outputs: the restatement sweeps the full matrix
    in stripes of 16 like sw/sw_inl.h, which decides LOCAL ties and the early exit"""
    g, w = dp_golden, sw_golden
    u = lambda v: int(np.int64(np.int32(np.uint32(v))))
    checked = 0
    for i in range(len(g["pat_off"]) - 1):
        p, t, _ = _case(g, i)
        for si, sw in enumerate(w["schemes"]):
            for bi, b in enumerate(w["bands"]):
                for typ in range(3):
                    want = tuple(int(v) for v in w["bsw"][i, si, bi, typ])
                    if want[0] < 0:
                        continue
                    ok, s, sk = orc.banded_sw(int(b), typ, sw, p, t)
                    assert (ok, s, u(sk[0]), u(sk[1])) == want, (i, si, b, typ)
                    checked += 1
            for blk in range(2):
                for typ in range(3):
                    for v, ms in enumerate((oracle.SCORE_MIN, int(g["min_scores"][i]))):
                        ok, s, sk = orc.full_sw(typ, blk, sw, p, t, ms)
                        assert (ok, s, u(sk[0]), u(sk[1])) == tuple(int(x) for x in w["fsw"][i, si, blk, typ, v]), (i, si, blk, typ, v)
                        checked += 1
        for blk in range(2):
            for typ in range(3):
                for v, ms in enumerate((oracle.SCORE_MIN, max(int(g["min_scores"][i]), -(len(p) // 4) - 1))):
                    ok, s, sk = orc.full_sw(typ, blk, oracle.ED_SW, p, t, ms)
                    assert (ok, s, u(sk[0]), u(sk[1])) == tuple(int(x) for x in w["fed"][i, blk, typ, v]), (i, blk, typ, v)
                    checked += 1
    assert checked > 20000


def test_hamming_backtrack_golden(orc, fm_golden, bt_golden):
    """nvbio::hamming_backtrack (approximate FM-index search, the reference's benchmark kernel): in `quirks` mode the restatement
    equals the reference's own code run over PackedStream patterns -- count, number of ranges, the ranges in delegate order --
    including what that code does at the start of the pattern (no stop at l == 0); in the default mode it equals a brute-force
    Hamming scan of the text (seed part exact, at most `mismatches` elsewhere)"""
    g, b = fm_golden, bt_golden
    hidx = oracle.HostIndex(len(g["text"]), int(g["primary"]), g["L2"], g["bwt_occ"], g["ssa"])
    stream, offs, text = b["stream"], b["offs"], g["text"]
    differ = 0
    for mi, (seed, mm) in enumerate(b["modes"]):
        for i in range(len(offs) - 1):
            L = int(offs[i + 1] - offs[i])
            c, n, rg = orc.hamming_backtrack(hidx, stream, int(offs[i]), L, int(seed), int(mm), quirks=True, cap=48)
            assert (c, n) == (int(b["counts"][mi, i]), int(b["n_ranges"][mi, i])), (mi, i)
            assert np.array_equal(rg.astype(np.int64), b["ranges"][mi, i, :min(n, 48)]), (mi, i)
            c2, _, _ = orc.hamming_backtrack(hidx, stream, int(offs[i]), L, int(seed), int(mm))
            mis = np.lib.stride_tricks.sliding_window_view(text, L) != stream[offs[i]:offs[i + 1]][None, :]
            hit = (mis.sum(1) == 0) if int(mm) == 0 else ((mis[:, L - int(seed):].sum(1) == 0) & (mis.sum(1) <= int(mm)))
            assert c2 == int(hit.sum()), (mi, i)
            differ += c != c2
    assert differ > 100            # the reference's fall-through at l == 0 inflates the counts of most hitting queries


def test_best2_sink_golden(orc, dp_golden, best2_golden):
    """aln::Best2Sink<int32>(distinct_dist) (sink.h:96-116): the best two distinct alignments of the banded and the full-matrix
    DP, on the reference's own outputs -- an order-dependent sink, so this also pins the order of the reports"""
    g, w = dp_golden, best2_golden
    S = len(g["schemes"])
    checked = seconds = 0
    for i in range(0, len(g["pat_off"]) - 1):
        sc = oracle.Scheme(*[int(v) for v in g["schemes"][i % S]])
        p, t, q = _case(g, i)
        for di, dist in enumerate(w["dists"]):
            for bi, b in enumerate(w["bands"]):
                for typ in range(3):
                    want = tuple(int(v) for v in w["banded"][i, di, bi, typ])
                    if want[0] < 0:
                        continue
                    ok, out = orc.banded_gotoh_best2(int(b), typ, sc, p, t, q, int(dist))
                    assert (ok,) + out == want, (i, dist, b, typ)
                    checked += 1; seconds += out[3] > oracle.SCORE_MIN
            for blk in range(2):
                for typ in range(3):
                    for v, ms in enumerate((oracle.SCORE_MIN, int(g["min_scores"][i]))):
                        ok, out = orc.full_gotoh_best2(typ, blk, sc, p, t, q, ms, int(dist))
                        assert (ok,) + out == tuple(int(x) for x in w["full"][i, di, blk, typ, v]), (i, dist, blk, typ, v)
                        checked += 1; seconds += out[3] > oracle.SCORE_MIN
    assert checked > 15000 and seconds > 5000


def test_full_traceback_golden(orc, dp_golden, ftb_golden):
    """score, source, sink and run-length CIGAR of the reference's full-matrix alignment_traceback"""
    g, t = dp_golden, ftb_golden
    n = len(g["pat_off"]) - 1
    u = lambda v: int(np.uint32(np.int64(v) & 0xFFFFFFFF))
    traced = 0
    for i in range(n):
        sc = _tb_scheme(g, i)
        p, x, q = _case(g, i)
        for typ in range(3):
            for v, ms in enumerate((oracle.SCORE_MIN, int(g["min_scores"][i]))):
                want = t["aln"][i, typ, v]
                ok, s, src, snk, cig = orc.full_gotoh_traceback(typ, sc, p, x, q, ms)
                assert (ok, s) == (int(want[0]), int(want[1])), (i, typ, v)
                assert src == (u(want[2]), u(want[3])) and snk == (u(want[4]), u(want[5])), (i, typ, v)
                lo, hi = t["cig_off"][i, typ, v]
                assert np.array_equal(cig, t["cigars"][lo:hi]), (i, typ, v, oracle.cigar_string(cig))
                if ok:      # the CIGAR spans the pattern and the source/sink text interval
                    ln = cig >> 2; ty = cig & 3
                    assert int(ln[(ty == 0) | (ty == 1) | (ty == 3)].sum()) == len(p)
                    assert int(ln[(ty == 0) | (ty == 2)].sum()) == snk[0] - src[0]
                traced += ok
    assert traced > 1500
    # the reference's functional tests (alignment_test.cu:741-755,816-828)
    p, x, _ = _case(g, 0)
    sc = oracle.Scheme.simple(2, -1, -1, -1)
    assert oracle.cigar_string(orc.full_gotoh_traceback(oracle.GLOBAL, sc, p, x)[4]) == "1M2D3M1D3M10D"
    assert oracle.cigar_string(orc.full_gotoh_traceback(oracle.LOCAL, sc, p, x)[4]) == "4M1D3M"
    assert oracle.cigar_string(orc.full_gotoh_traceback(oracle.SEMI_GLOBAL, sc, p, x)[4]) == "4M1D3M"
    p, x, _ = _case(g, 2)
    assert oracle.cigar_string(orc.full_gotoh_traceback(oracle.SEMI_GLOBAL, oracle.Scheme.simple(0, -5, -8, -3), p, x)[4]) == "6I138M"


def test_full_gotoh_golden(orc, dp_golden):
    g = dp_golden
    S = len(g["schemes"])
    n = len(g["pat_off"]) - 1
    for i in range(n):
        sc = oracle.Scheme(*[int(v) for v in g["schemes"][i % S]])
        p, t, q = _case(g, i)
        for blk in range(2):
            for typ in range(3):
                for v, ms in enumerate((oracle.SCORE_MIN, int(g["min_scores"][i]))):
                    ok, s, sk = orc.full_gotoh(typ, blk, sc, p, t, q, ms)
                    want = g["full"][i, blk, typ, v]
                    assert (ok, s, sk[0], sk[1]) == tuple(int(x) for x in want), (i, blk, typ, v)


def test_known_answers_of_the_reference_tests(orc, dp_golden):
    """nvbio-test/alignment_test.cu:709-828: '4M1D3M' (banded 7), '147M2D3M' (banded 31), '6I138M'"""
    g = dp_golden
    for k in range(int(g["n_known"])):
        sc = oracle.Scheme(*[int(v) for v in g["known_schemes"][k]])
        p, t, _ = _case(g, k)
        for typ in range(3):
            ok, s, sk = orc.banded_gotoh(7 if k == 0 else 31, typ, sc, p, t)
            assert (ok, s, sk[0], sk[1]) == tuple(int(v) for v in g["known"][k, 0, typ])
            for blk in range(2):
                ok, s, sk = orc.full_gotoh(typ, blk, sc, p, t)
                assert (ok, s, sk[0], sk[1]) == tuple(int(v) for v in g["known"][k, 1 + blk, typ])
    # 150 bp read, band 31, Gotoh(0,-5,-8,-3), semi-global -> 147M2D3M: one gap of 2 = -8-3
    sc = oracle.Scheme.simple(0, -5, -8, -3)
    p, t, _ = _case(g, 1)
    assert orc.banded_gotoh(31, oracle.SEMI_GLOBAL, sc, p, t) == (1, -11, (165, 150))
    # 7x20 pair, Gotoh(2,-1,-1,-1), full semi-global and local -> 4M1D3M: 7 matches and one gap of 1
    sc = oracle.Scheme.simple(2, -1, -1, -1)
    p, t, _ = _case(g, 0)
    for blk in range(2):
        assert orc.full_gotoh(oracle.SEMI_GLOBAL, blk, sc, p, t)[1] == 13
        assert orc.full_gotoh(oracle.LOCAL, blk, sc, p, t)[1] == 13


def test_mismatch_quality_ramp(orc):
    # QualCost (nvBowtie/bowtie2/cuda/scoring.h:84-88): min + int(min(q,40)/40 * (max-min))
    sc = oracle.Scheme(2, 2, 6, -8, -3, -8, -3)
    assert [orc.mismatch(sc, q) for q in (0, 9, 10, 20, 30, 39, 40, 63, 255)] == [-2, -2, -3, -4, -5, -5, -6, -6, -6]


def test_band31_out_of_range_text_reads_as_T(orc):
    """past the text end the band-31 cache (2-bit packed) turns the 255
    sentinel into 3 ('T') on later rows; LOCAL scores equal T-padding to 181"""
    rng = np.random.default_rng(5)
    sc = oracle.Scheme(2, 6, 6, -8, -3, -8, -3)
    for _ in range(200):
        N = int(rng.integers(150, 180))
        t = rng.integers(0, 4, N, dtype=np.uint8)
        p = t[:150].copy()
        p[rng.integers(0, 150, 4)] = rng.integers(0, 4, 4)
        a = orc.banded_gotoh(31, oracle.LOCAL, sc, p, t)
        tp = np.concatenate([t, np.full(181 - N, 3, dtype=np.uint8)])
        b = orc.banded_gotoh(31, oracle.LOCAL, sc, p, tp)
        assert a[1] == b[1]


def test_score_reduce_in_descending_order_is_best_and_best_distinct(orc):
    """nvBowtie's score_reduce loop (reduce_inl.h:65-140) fed a read's candidates in descending order of the selection key
    ends with a1 = the largest key and a2 = the largest key among the candidates `distinct` from a1 that beat the threshold --
    the order-free form the GPU passes compute (parity unpinned: restated from device-only code); plus a few fixed points of
    BowtieMapq2 / BowtieMapq3 (mapq.h)"""
    rng = np.random.default_rng(9)
    for _ in range(300):
        n = int(rng.integers(1, 12))
        scores = rng.integers(-120, 1, n).astype(np.int32)
        pos = (1000 + rng.integers(0, 4, n) * rng.integers(1, 200) + rng.integers(0, 3, n)).astype(np.uint32)
        rc = rng.integers(0, 2, n).astype(np.uint8)
        key = ((scores.astype(np.int64) + (1 << 20)) << 34) | (rc.astype(np.int64) << 33) | pos
        key = np.unique(key)[::-1]                                   # one entry per (score, strand, position), descending
        s = ((key >> 34) - (1 << 20)).astype(np.int32); p = (key & ((1 << 33) - 1)).astype(np.uint32); r = ((key >> 33) & 1).astype(np.uint8)
        worst = -91
        out = orc.score_reduce(s, p, r, 150, worst)
        if s[0] <= worst:
            assert out[0] == 0 and out[4] == 0
            continue
        assert out[:4] == (1, int(s[0]), int(p[0]), int(r[0]))
        far = (r != r[0]) | ~((p[0].astype(np.int64) >= p.astype(np.int64) - np.minimum(p, 75)) & (p[0].astype(np.int64) <= p.astype(np.int64) + 75))
        cand = np.nonzero(far & (s > worst) & ~((p == p[0]) & (r == r[0])))[0]
        if len(cand) == 0:
            assert out[4] == 0
        else:
            k = cand[0]
            assert out[4:] == (1, int(s[k]), int(p[k]), int(r[k]))
    # end-to-end, 150 bp: perfect unique read; read at the threshold; two equal alignments
    assert orc.mapq(2, True, 0, -90, 0, False, 0) == 42 and orc.mapq(3, True, 0, -90, 0, False, 0) == 44
    assert orc.mapq(2, True, 0, -90, -90, False, 0) == 0 and orc.mapq(2, True, 0, -90, -91, False, 0) == 0
    assert orc.mapq(2, True, 0, -90, -6, True, -6) == 1 and orc.mapq(2, True, 0, -90, 0, True, -90) == 39
    assert orc.mapq(2, False, 300, 50, 300, False, 0) == 44 and orc.mapq(2, False, 300, 50, 300, True, 300) == 1


def test_seed_hit_deque_golden(orc, deque_golden):
    """nvBowtie's per-read seed-hit deque: the oracle's restatement of the interval heap leaves the array exactly as the reference's
    own priority_deque does (nvbio/basic/priority_deque.h, driven by oracle/ref ref_hit_deque_run) after 600 operation sequences
    -- pushes under the max_hits rule, pop_top / pop_bottom, select's in-place row pops -- and returns the same rows"""
    g = deque_golden
    for k in range(len(g["max_hits"])):
        a, b = int(g["off"][k]), int(g["off"][k + 1])
        heap, rows = orc.hit_deque_run(g["ops"][a:b], g["begins"][a:b], g["bits"][a:b], int(g["max_hits"][k]))
        assert np.array_equal(heap, g["heaps"][int(g["hoff"][k]):int(g["hoff"][k + 1])]), k
        assert np.array_equal(rows, g["rows"][a:b]), k


def test_exact_mapper_select_and_effort_rules(orc):
    """the bookkeeping around the deque, restated from device-only code (parity unpinned beyond the container): the exact mapper keeps
    the max_hits smallest ranges and asks for reseeding on repetitive reads; select walks the ranges smallest first, row by row;
    the effort counter stops a read after max_effort failures in a row past min_ext"""
    # 5 seeds: fw ranges of sizes 1, 3, -, 50, 1 and rc ranges -, 2, -, -, 1000 (inclusive); max_hits = 4
    fw = np.array([[10, 10], [20, 22], [1, 0], [100, 149], [7, 7]], dtype=np.uint32)
    rc = np.array([[1, 0], [30, 31], [1, 0], [1, 0], [1000, 1999]], dtype=np.uint32)
    off = np.arange(5) * 15
    deque, reseed = orc.map_exact_read(fw, rc, off, 150, 22, 4, 1000)
    sizes = sorted(int(b & 0xFFFFF) for b in deque[:, 1])
    # a full deque drops its LARGEST range before every push, whatever the newcomer's size (mapping_inl.h:242-244): the 50-row range
    # goes when the fifth hit arrives, the 3-row one when the 1000-row range arrives last -- and that one stays
    assert sizes == [1, 1, 2, 1000] and not reseed                    # mean range size 1057 / 6 < rep_seeds
    assert orc.map_exact_read(fw, rc, off, 150, 22, 4, 100)[1]        # rep_seeds = 100: 1057 >= 100 * 6 -> reseed
    assert orc.map_exact_read(fw * 0 + [1, 0], rc * 0 + [1, 0], off, 150, 22, 4, 1000)[1]     # no hit at all -> reseed
    # positions: forward hits carry read_len - offset - seed_len, reverse-complemented ones the offset; rc in bit 30
    for begin, bits in deque:
        pos, is_rc = (int(bits) >> 20) & 0x3FF, (int(bits) >> 30) & 1
        j = [k for k in range(5) if (rc if is_rc else fw)[k, 0] == begin][0]
        assert pos == (off[j] if is_rc else 150 - off[j] - 22)
    # select: rows come out smallest range first, each range front to back; the top flag drops once the first range is exhausted
    rows, flags, top = [], [], 1
    for _ in range(8):
        ok, row, seed, top, deque = orc.select_read(deque, top)
        assert ok
        rows.append(row); flags.append((seed >> 14) & 1)
    assert set(rows[:2]) == {7, 10} and rows[2:4] == [30, 31] and rows[4:] == [1000, 1001, 1002, 1003]
    assert flags[0] == 1 and flags[1:] == [0] * 7
    one = np.array([[5, 2 | (3 << 20)]], dtype=np.uint32)              # a single two-row hit: two rows, then the read leaves the queue
    ok, row, _, _, one = orc.select_read(one, 1); assert ok and row == 5
    ok, row, _, _, one = orc.select_read(one, 1); assert ok and row == 6
    assert not orc.select_read(one, 1)[0]
    # effort: worst score -91; a hit improves the best, a distinct worse one becomes second, repeats of either are skipped for free,
    # failures past min_ext count down from max_effort
    best, trys = [-91, 0xFFFFFFFF, 0, -91, 0xFFFFFFFF, 0], 3
    best, trys, erase = orc.score_reduce_effort(best, trys, -12, 5000, 0, 0, 150, 0, 3, 2, 400)
    assert best[:3] == [-12, 5000, 0] and trys == 3 and not erase
    best, trys, erase = orc.score_reduce_effort(best, trys, -30, 9000, 1, 0, 150, 1, 3, 2, 400)
    assert best[3:] == [-30, 9000, 1] and trys == 3
    best, trys, erase = orc.score_reduce_effort(best, trys, -50, 5000, 0, 0, 150, 2, 3, 2, 400)      # the best locus again: skipped
    assert trys == 3 and not erase
    best, trys, erase = orc.score_reduce_effort(best, trys, -50, 5040, 0, 0, 150, 1, 3, 2, 400)      # a failure before min_ext: not counted
    assert trys == 3 and not erase
    for k, want in ((2, 2), (3, 1)):
        best, trys, erase = orc.score_reduce_effort(best, trys, -50, 5040, 0, 0, 150, k, 3, 2, 400)
        assert trys == want and not erase
    best, trys, erase = orc.score_reduce_effort(best, trys, -50, 5040, 0, 0, 150, 4, 3, 2, 400)
    assert trys == 0 and erase                                        # third failure in a row: the read is done
    best, trys, erase = orc.score_reduce_effort(best, 2, -50, 5040, 0, 1, 150, 4, 3, 2, 400)        # top-seed hits never count
    assert trys == 2 and not erase
    assert orc.score_reduce_effort(best, 2, -50, 5040, 0, 1, 150, 400, 3, 2, 400)[2]               # max_ext reached: stop regardless


def test_generic_rank_dictionary_golden(orc, rankdict_golden):
    """the generic rank dictionary (plain 32- / 64-bit words, separate occurrence table, K = 64 / 128, 32- / 64-bit indices) against the
    reference's own dispatch_rank in the two configurations its test runs (rank_test.cu:83-227): occ table, totals, rank of every
    (i, c) including i = -1"""
    import importlib.util, os
    spec = importlib.util.spec_from_file_location("mg", os.path.join(os.path.dirname(__file__), "golden", "make_golden.py"))
    mg = importlib.util.module_from_spec(spec); spec.loader.exec_module(mg)
    g = rankdict_golden
    sym = g["sym"]; n = len(sym)
    for wb, K in ((32, 64), (64, 128)):
        tw = mg.pack_words(sym, wb)
        minus1 = (1 << wb) - 1
        idx = np.repeat(np.concatenate([np.arange(n, dtype=np.uint64), np.array([minus1], dtype=np.uint64)]), 4)
        cs = np.tile(np.arange(4, dtype=np.uint8), n + 1)
        occ, cnt, r = orc.rank_generic(tw, wb, n, K, wb, idx, cs)
        assert np.array_equal(occ.astype(np.uint64), g["occ%d" % wb]) and np.array_equal(cnt, g["cnt%d" % wb])
        assert np.array_equal(r.reshape(n + 1, 4), g["rank%d" % wb])
        assert np.array_equal(g["rank4_%d" % wb], g["rank%d" % wb][:n])          # the reference's rank4 agrees with its rank
        run = np.cumsum(sym[:, None] == np.arange(4)[None, :], axis=0)           # and both with running counts (rank_test.cu:46-79)
        assert np.array_equal(g["rank%d" % wb][:n], run.astype(np.uint64))


def test_arrival_order_rule_differs_from_the_order_free_second_best_on_near_ties(orc):
    """documented deviation: nvBowtie's score_reduce demotes the old best to second best WITHOUT a distinctness
    test (reduce_inl.h:113-114), so its second alignment depends on the order hits arrive in; the default pipeline's order-free pass
    (nvbio_second_candidate_reduce) keeps the best candidate that is distinct from the final best.  Two hits 10 bp apart on one strand:
    arriving worse-first the reference keeps both (a2 = the worse, not distinct); arriving better-first it keeps only the best; the
    order-free rule says "no second alignment" either way.  nvbio_score_reduce_effort / pipeline.nvbowtie_best_approx implement the
    reference's rule."""
    worst = -90
    a = orc.score_reduce(np.array([-12, -6], dtype=np.int32), np.array([5000, 5010], dtype=np.uint32), np.array([0, 0], dtype=np.uint8), 150, worst)
    b = orc.score_reduce(np.array([-6, -12], dtype=np.int32), np.array([5010, 5000], dtype=np.uint32), np.array([0, 0], dtype=np.uint8), 150, worst)
    assert a[:4] == (1, -6, 5010, 0) and a[4:] == (1, -12, 5000, 0)        # worse first: the demoted old best stays as a2
    assert b[:4] == (1, -6, 5010, 0) and b[4] == 0                          # better first: the near-by worse hit is not distinct -> no a2
    # the order-free rule = the reference's loop over candidates in DESCENDING key order = case b, whatever the arrival order


def test_myers_golden(orc, myers_golden):
    """the Myers bit-vector aligner (myers_banded_inl.h:172-315, fmmap's aligner) against the reference's own outputs: 1,500 cases over
    bands 3 / 7 / 15 / 31, GLOBAL and SEMI_GLOBAL, min_score on both sides of the reference's int16 truncation"""
    g = myers_golden
    for k in range(len(g["band"])):
        pat = g["pats"][g["pat_off"][k]:g["pat_off"][k + 1]]; txt = g["txts"][g["txt_off"][k]:g["txt_off"][k + 1]]
        ok, sc, sk = orc.banded_myers(int(g["band"][k]), int(g["typ"][k]), pat, txt, int(g["min_score"][k]))
        assert (ok, sc, sk[0], sk[1]) == tuple(int(v) for v in g["out"][k]), k


def test_staged_banded_golden(orc, staged_golden):
    """the staged scheduler's windowed banded scoring (batched_stream.h:117-285 over gotoh_banded_inl.h:703-727) against the
    reference's own outputs: 3,000 cases, 0..9 windows of 32 rows, all three alignment types, min_score on both sides of what the
    pair can reach, a scheme that drives scores below the int16 checkpoint clamp"""
    g = staged_golden
    early = 0
    for k in range(len(g["band"])):
        pat = g["pats"][g["pat_off"][k]:g["pat_off"][k + 1]]; txt = g["txts"][g["txt_off"][k]:g["txt_off"][k + 1]]
        q = g["quals"][g["pat_off"][k]:g["pat_off"][k + 1]]
        q = None if (len(q) and q[0] == 255) else q
        ok, sc, sk = orc.banded_gotoh_staged(int(g["band"][k]), int(g["typ"][k]), oracle.Scheme(*[int(v) for v in g["schemes"][g["scheme"][k]]]),
                                             pat, txt, int(g["min_score"][k]), q)
        assert (ok, sc, sk[0], sk[1]) == tuple(int(v) for v in g["out"][k][:4]), k
        early += ok == 0
    assert early > 500


def test_banded_sw_traceback_golden(orc, swtb_golden):
    """traceback of the linear-gap Smith-Waterman aligner (banded_inl.h:354-417 over sw/sw_banded_inl.h; unequal deletion / insertion
    costs, the edit-distance scheme; its LOCAL walk that never meets a SINK) against the reference's own outputs: 2,500 cases"""
    g = swtb_golden
    for k in range(len(g["band"])):
        pat = g["pats"][g["pat_off"][k]:g["pat_off"][k + 1]]; txt = g["txts"][g["txt_off"][k]:g["txt_off"][k + 1]]
        ok, sc, src, snk, cig = orc.banded_sw_traceback(int(g["band"][k]), int(g["typ"][k]), g["schemes"][g["scheme"][k]], pat, txt)
        want = g["out"][k]
        assert ok == want[0] and sc == want[1], k
        if ok:
            assert src == (want[2], want[3]) and snk == (want[4], want[5]), k
            assert np.array_equal(cig, g["cigars"][g["cig_off"][k]:g["cig_off"][k + 1]]), k
    assert int(g["out"][:, 0].sum()) > 2000


def test_full_sw_traceback_golden(orc, fswtb_golden):
    """full-matrix traceback of the linear-gap Smith-Waterman aligner (alignment_inl.h:355-455 over sw/sw_inl.h:306-392,1476-1694; unequal
    deletion / insertion costs, the edit-distance scheme, SINK cells of LOCAL walks, the stripe early exit) against the reference's own
    outputs: 2,000 cases"""
    g = fswtb_golden
    for k in range(len(g["typ"])):
        pat = g["pats"][g["pat_off"][k]:g["pat_off"][k + 1]]; txt = g["txts"][g["txt_off"][k]:g["txt_off"][k + 1]]
        ok, sc, src, snk, cig = orc.full_sw_traceback(int(g["typ"][k]), g["schemes"][g["scheme"][k]], pat, txt, int(g["min_score"][k]))
        want = g["out"][k]
        assert ok == want[0] and sc == want[1], k
        if ok:
            assert src == (want[2], want[3]) and snk == (want[4], want[5]), k
            assert np.array_equal(cig, g["cigars"][g["cig_off"][k]:g["cig_off"][k + 1]]), k
    assert 1900 < int(g["out"][:, 0].sum()) < 2000

