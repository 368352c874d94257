#!/usr/bin/env python3
"""Generate the golden fixtures in tests/golden/ from the REFERENCE's own host code.

Runs only in the development container (needs /root/reference and oracle/_ref/libnvbio_ref.so,
built by `make -C oracle ref`).  The fixtures are data: seeded inputs plus the outputs the
reference computed for them.  Nothing from the reference's sources is stored.

    python tests/golden/make_golden.py

Fixtures:
  fm_golden.npz   a 3,000-symbol seeded text; the index the reference built for it (sais SA, BWT,
                  occ, interleaved bwt_occ, SSA); rank/rank4 for every row; match ranges for 400
                  queries (hits, misses, N's; forward and reverse scans); locate for 600 rows.
  dp_golden.npz   the known-answer pairs of nvbio-test/alignment_test.cu:709-828 and 400 seeded
                  random pairs (substitutions, indels, N's, qualities, clipped windows, N < M),
                  each scored by the reference with banded Gotoh (bands 3/7/15/31 x 3 types) and
                  full-matrix Gotoh (pattern/text blocking x 3 types, with and without min_score).
  ed_golden.npz   the same pairs scored by the reference's banded edit-distance aligner.
  ftb_golden.npz  the same pairs traced back through the full matrix (alignment_traceback).
  bt_golden.npz   hamming_backtrack (approximate FM-index search) over that index.
  best2_golden.npz the same pairs scored into aln::Best2Sink (best two distinct alignments).
  sw_golden.npz   the same pairs scored by the reference's linear-gap Smith-Waterman aligner (banded and full
                  matrix) and by its full-matrix edit-distance aligner.
  tb_golden.npz   the same pairs traced back by the reference (banded_alignment_traceback, bands
                  3/7/15/31 x 3 types): score, source, sink and the run-length CIGAR.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import oracle  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))

DNA = {"A": 0, "C": 1, "G": 2, "T": 3}


def enc(s):
    return np.array([DNA[c] for c in s], dtype=np.uint8)


# input strings of the reference's own functional tests (nvbio-test/alignment_test.cu:721-722,768-769,801-807)
KNOWN = [
    ("ACAACTA", "AAACACCCTAACACACTAAA"),
    ("TTATGTAGGTGGTCTGGTTTTTGCCTTTTAAGCTTCTGCAAAAAACAACAACAAACTTGTGGTATTACACTGACTCTACAGATCAATTTGGGGACAACTTCCATG"
     "TGTTCCACCACCAATACTGAATCTTTCAATCGACTGACGTGGTAT",
     "ATCGGATTCTTTCTTACTTGTAGGTGGTCTGGTTTTTGCCTTTTAAGCTTCTGCAAAAAACAACAACAAACTTGTGGTATTACACTGACTCTACAGATCAATTTG"
     "GGGACAACTTCCATGTGTTCCACCACCAATACTGAATCTTTCAATCGACTGACGTGGTATCTCTCTCTCCATCTAT"),
    ("TAGGAGGTAACATGTATGGAGCATTTACCATAGGCCAAGCACTGTTCTAAGAACTTCGGACATGTTATCTCACTTGTATAAGTACTTAGGTGCCTACAACATAAG"
     "CAGCACCTGGTAAATTAAGTATTGAAAAAATGCAGATCG",
     "CAGCACTGACCGGTGAGCATAAACCCTGGGGATGCCCAGAGCTGGTACAGCCAGGAGCTCCAGAAGCGTGGGATTCTCAGAGGGAAGTGGAGCTCACTGCTCTAC"
     "AGGTCCTATTCAAGTTAGAAAGTAAGATACAATGCACACAAAGCCAAATTGTCATCATTCAGCTCCTATTACAGGGGAACTAAGAGCTGCATTGAAAATTATTTG"
     "CAAAGCTTGTAAGTGGTTCTGCCACTTATTAGCCGTGTGAACCTTAGCAAATTACCTAGCGTCTCTGAGTTTCAACTTCCTCATCTACAAAATAGAAATGATAAT"
     "AATAACCGCATCGCAAGAGTTGTTGGAAAAATGAAAATGAGGTATCATAGGAGGTAACATGTATGGAGCATTTACCATAGGCCAAGCACTGTTCTAAGAACTTCG"
     "GACATGTTATCTCACTTGTATAAGTACTTAGGTGCCTACAACATAAACAGCACCTGGTAAATTAAGTATTGAAAAAATGC"),
]

# scoring schemes: the reference tests' SimpleGotohScheme settings (alignment_test.cu:743-747,771-775),
# sw-benchmark's (2,-1,-2,-1), nvBowtie's local() and default (e2e) schemes
# (nvBowtie/bowtie2/cuda/scoring_inl.h:72-114) and one with asymmetric gap costs
SCHEMES = [
    (2, 1, 1, -1, -1, -1, -1),
    (0, 5, 5, -8, -3, -8, -3),
    (2, 1, 1, -2, -1, -2, -1),
    (2, 2, 6, -8, -3, -8, -3),
    (0, 2, 6, -8, -3, -8, -3),
    (1, 3, 3, -15, -4, -11, -2),
]


def make_fm(R):
    rng = np.random.default_rng(20240607)
    n = 3000
    text = rng.integers(0, 4, n, dtype=np.uint8)
    text[1000:1400] = np.tile(np.array([0, 1, 0, 2], dtype=np.uint8), 100)      # a repeat: wide SA ranges
    idx = R.build_index(text)

    ranks = np.zeros((n + 2, 4), dtype=np.uint32)
    ranks4 = np.zeros((n + 1, 4), dtype=np.uint32)
    for k in range(-1, n + 1):
        for c in range(4):
            ranks[k + 1, c] = R.rank(idx, k, c)
        if k >= 0:
            ranks4[k] = R.rank4(idx, k)
    pairs = np.zeros((500, 3), dtype=np.int64)
    ranks2 = np.zeros((500, 2), dtype=np.uint32)
    for i in range(500):
        l = int(rng.integers(-1, n + 1))
        r = int(rng.integers(max(l, 0), n + 1))
        c = int(rng.integers(0, 4))
        pairs[i] = (l, r, c)
        ranks2[i] = R.rank2(idx, l, r, c)

    Q = 400
    lens = rng.integers(1, 33, Q)
    lens[:40] = 22
    offs = np.zeros(Q + 1, dtype=np.uint32)
    offs[1:] = np.cumsum(lens)
    syms = rng.integers(0, 4, int(offs[-1]), dtype=np.uint8)
    for q in range(0, Q, 2):
        p = int(rng.integers(0, n - lens[q] + 1))
        syms[offs[q]:offs[q + 1]] = text[p:p + lens[q]]
    syms[rng.integers(0, len(syms), 25)] = 4
    ranges_bwd = R.match_batch(idx, syms, offs, False)
    ranges_fwd = R.match_batch(idx, syms, offs, True)

    rows = np.concatenate([np.arange(0, 200), rng.integers(0, n + 1, 398), [idx.primary, n]]).astype(np.uint32)
    pos = R.locate_batch(idx, rows)
    jt = R.locate_ssa_batch(idx, rows)

    np.savez_compressed(
        os.path.join(HERE, "fm_golden.npz"),
        text=text, sa=idx.sa, primary=np.uint32(idx.primary), L2=idx.L2, bwt_occ=idx.bwt_occ, ssa=idx.ssa,
        ranks=ranks, ranks4=ranks4, rank2_args=pairs, rank2=ranks2,
        q_syms=syms, q_offs=offs, ranges_bwd=ranges_bwd, ranges_fwd=ranges_fwd,
        rows=rows, pos=pos, jt=jt)
    R.destroy(idx)
    print("fm_golden.npz: n=%d, %d queries, %d rows" % (n, Q, len(rows)))


def make_dp(R):
    rng = np.random.default_rng(977)
    pats, txts, quals = [], [], []
    for p, t in KNOWN:
        pats.append(enc(p)); txts.append(enc(t)); quals.append(None)
    for it in range(400):
        M = int(rng.integers(1, 161)) if it % 10 else 150
        kind = it % 8
        if kind == 0:       # window clipped at the text end / shorter than the pattern
            N = int(rng.integers(max(1, M - 3), M + 12))
        elif kind == 1:     # long text for full DP
            N = int(rng.integers(M, M + 350))
        else:
            N = M + 31
        N = max(N, 30)      # the banded reference reads text[0..BAND-2] unconditionally
        txt = rng.integers(0, 4, N, dtype=np.uint8)
        st = int(rng.integers(0, max(1, min(31, N - M + 1)))) if N >= M else 0
        pat = txt[st:st + M].copy()
        if len(pat) < M:
            pat = np.concatenate([pat, rng.integers(0, 4, M - len(pat), dtype=np.uint8)])
        mut = rng.random(M) < 0.05
        pat[mut] = rng.integers(0, 5, int(mut.sum()))
        if it % 3 == 0 and M > 12:
            k = int(rng.integers(2, M - 6)); g = int(rng.integers(1, 4))
            if it % 2:
                pat = np.concatenate([pat[:k], pat[k + g:], rng.integers(0, 4, g, dtype=np.uint8)])
            else:
                pat = np.concatenate([pat[:k], rng.integers(0, 4, g, dtype=np.uint8), pat[k:M - g]])
        pats.append(pat); txts.append(txt)
        quals.append(rng.integers(0, 64, M, dtype=np.uint8) if it % 2 else None)

    n = len(pats)
    pat_off = np.zeros(n + 1, dtype=np.uint32); pat_off[1:] = np.cumsum([len(p) for p in pats])
    txt_off = np.zeros(n + 1, dtype=np.uint32); txt_off[1:] = np.cumsum([len(t) for t in txts])
    has_q = np.array([q is not None for q in quals], dtype=np.uint8)
    qual_flat = np.concatenate([q if q is not None else np.zeros(len(p), dtype=np.uint8) for q, p in zip(quals, pats)])
    bands = [3, 7, 15, 31]
    S = len(SCHEMES)
    # banded[case, band, type] = (ok, score, sink.x, sink.y); scheme = case % S
    banded = np.zeros((n, len(bands), 3, 4), dtype=np.int64)
    # full[case, blocking, type, min_score_variant] = (ok, score, sink.x, sink.y)
    full = np.zeros((n, 2, 3, 2, 4), dtype=np.int64)
    min_scores = np.full(n, oracle.SCORE_MIN, dtype=np.int32)
    for i in range(n):
        sc = oracle.Scheme(*SCHEMES[i % S])
        # a finite min_score exercises the stripe early exit (gotoh_inl.h:706-710,1106-1110)
        min_scores[i] = int(rng.integers(-40, 60)) if i % 3 else sc.match * len(pats[i]) - int(rng.integers(0, 30))
        for bi, b in enumerate(bands):
            for typ in range(3):
                ok, s, sk = R.banded_gotoh(b, typ, sc, pats[i], txts[i], quals[i])
                banded[i, bi, typ] = (ok, s, sk[0], sk[1])
        for blk in range(2):
            for typ in range(3):
                for v, ms in enumerate((oracle.SCORE_MIN, int(min_scores[i]))):
                    ok, s, sk = R.full_gotoh(typ, blk, sc, pats[i], txts[i], quals[i], ms)
                    full[i, blk, typ, v] = (ok, s, sk[0], sk[1])
    # the reference's functional tests, under the schemes those tests use (alignment_test.cu:743-828):
    # known[k, variant, type] with variant 0 = banded (band 7 for the 7x20 pair, 31 otherwise),
    # 1 = full pattern-blocking, 2 = full text-blocking
    known_scheme = [SCHEMES[0], SCHEMES[1], SCHEMES[1]]
    known = np.zeros((len(KNOWN), 3, 3, 4), dtype=np.int64)
    for k in range(len(KNOWN)):
        sc = oracle.Scheme(*known_scheme[k])
        for typ in range(3):
            ok, s_, sk = R.banded_gotoh(7 if k == 0 else 31, typ, sc, pats[k], txts[k], None, simple=True)
            known[k, 0, typ] = (ok, s_, sk[0], sk[1])
            for blk in range(2):
                ok, s_, sk = R.full_gotoh(typ, blk, sc, pats[k], txts[k], None)
                known[k, 1 + blk, typ] = (ok, s_, sk[0], sk[1])

    np.savez_compressed(
        os.path.join(HERE, "dp_golden.npz"),
        known=known, known_schemes=np.array(known_scheme, dtype=np.int32),
        pats=np.concatenate(pats), pat_off=pat_off, txts=np.concatenate(txts), txt_off=txt_off,
        quals=qual_flat, has_quals=has_q, schemes=np.array(SCHEMES, dtype=np.int32), bands=np.array(bands),
        banded=banded, full=full, min_scores=min_scores, n_known=np.int32(len(KNOWN)))
    print("dp_golden.npz: %d pairs" % n)


def make_tb(R):
    """tb_golden.npz: the reference's banded_alignment_traceback<BAND,1024,16> on every pair of
    dp_golden.npz (bands 3/7/15/31 x 3 types, scheme = case % S; the known-answer pairs under the
    schemes of alignment_test.cu:743-828): Alignment {score, source, sink} and the ops / clips the
    Backtracer received, folded into nvBowtie's run-length io::Cigar elements (backtracking order)."""
    g = np.load(os.path.join(HERE, "dp_golden.npz"))
    n = len(g["pat_off"]) - 1
    bands = [int(b) for b in g["bands"]]
    S = len(g["schemes"])
    nk = int(g["n_known"])
    aln = np.zeros((n, len(bands), 3, 6), dtype=np.int64)       # traced, score, source.x, source.y, sink.x, sink.y
    cig_off = np.zeros((n, len(bands), 3, 2), dtype=np.int64)   # [begin, end) into cigars
    cigars = []
    pos = 0
    for i in range(n):
        pat = g["pats"][g["pat_off"][i]:g["pat_off"][i + 1]]
        txt = g["txts"][g["txt_off"][i]:g["txt_off"][i + 1]]
        q = g["quals"][g["pat_off"][i]:g["pat_off"][i + 1]] if g["has_quals"][i] else None
        sv = g["known_schemes"][i] if i < nk else g["schemes"][i % S]
        sc = oracle.Scheme(*[int(x) for x in sv])
        for bi, b in enumerate(bands):
            for typ in range(3):
                if len(txt) < b - 1:        # the reference reads text[0..BAND-2] unconditionally: undefined, not pinned
                    aln[i, bi, typ, 0] = -1
                    cig_off[i, bi, typ] = (pos, pos)
                    continue
                r, s_, src, snk, ops, clips = R.banded_gotoh_traceback(b, typ, sc, pat, txt, q)
                traced = 1 if r == 2 else 0
                assert r in (0, 2)
                c = oracle.cigar_from_ops(ops, *clips) if traced else np.zeros(0, dtype=np.uint16)
                aln[i, bi, typ] = (traced, s_, np.int64(np.int32(np.uint32(src[0]))), np.int64(np.int32(np.uint32(src[1]))),
                                   np.int64(np.int32(np.uint32(snk[0]))), np.int64(np.int32(np.uint32(snk[1]))))
                cig_off[i, bi, typ] = (pos, pos + len(c)); pos += len(c)
                cigars.append(c)
    np.savez_compressed(os.path.join(HERE, "tb_golden.npz"), aln=aln, cig_off=cig_off,
                        cigars=np.concatenate(cigars).astype(np.uint16), bands=np.array(bands))
    print("tb_golden.npz: %d pairs, %d cigar elements" % (n, pos))


def make_ftb(R):
    """ftb_golden.npz: the reference's full-matrix alignment_traceback<256,1024,64> on every pair of dp_golden.npz,
    3 types x {no, finite} min_score (scheme = case % S; known-answer pairs under their own schemes): Alignment and
    the run-length CIGAR (x = text, y = pattern)"""
    g = np.load(os.path.join(HERE, "dp_golden.npz"))
    n = len(g["pat_off"]) - 1
    S = len(g["schemes"]); nk = int(g["n_known"])
    aln = np.zeros((n, 3, 2, 6), dtype=np.int64)
    cig_off = np.zeros((n, 3, 2, 2), dtype=np.int64)
    cigars = []; pos = 0
    for i in range(n):
        pat = g["pats"][g["pat_off"][i]:g["pat_off"][i + 1]]
        txt = g["txts"][g["txt_off"][i]:g["txt_off"][i + 1]]
        q = g["quals"][g["pat_off"][i]:g["pat_off"][i + 1]] if g["has_quals"][i] else None
        sv = g["known_schemes"][i] if i < nk else g["schemes"][i % S]
        sc = oracle.Scheme(*[int(x) for x in sv])
        for typ in range(3):
            for v, ms in enumerate((oracle.SCORE_MIN, int(g["min_scores"][i]))):
                r, s_, src, snk, ops, clips = R.full_gotoh_traceback(typ, sc, pat, txt, q, ms)
                traced = 1 if r == 2 else 0
                assert r in (0, 2)
                c = oracle.cigar_from_ops(ops, *clips) if traced else np.zeros(0, dtype=np.uint16)
                aln[i, typ, v] = (traced, s_, np.int64(np.int32(np.uint32(src[0]))), np.int64(np.int32(np.uint32(src[1]))),
                                  np.int64(np.int32(np.uint32(snk[0]))), np.int64(np.int32(np.uint32(snk[1]))))
                cig_off[i, typ, v] = (pos, pos + len(c)); pos += len(c)
                cigars.append(c)
    np.savez_compressed(os.path.join(HERE, "ftb_golden.npz"), aln=aln, cig_off=cig_off, cigars=np.concatenate(cigars).astype(np.uint16))
    print("ftb_golden.npz: %d pairs, %d cigar elements" % (n, pos))


def make_ed(R):
    """ed_golden.npz: the reference's banded edit-distance aligner (EditDistanceAligner<TYPE>, the aligner of
    examples/fmmap and of nvBowtie --scoring ed) on every pair of dp_golden.npz, bands 3/7/15/31 x 3 types"""
    g = np.load(os.path.join(HERE, "dp_golden.npz"))
    n = len(g["pat_off"]) - 1
    bands = [int(b) for b in g["bands"]]
    ed = np.zeros((n, len(bands), 3, 4), dtype=np.int64)          # ok, score, sink.x, sink.y  (ok = -1: not pinned)
    for i in range(n):
        pat = g["pats"][g["pat_off"][i]:g["pat_off"][i + 1]]
        txt = g["txts"][g["txt_off"][i]:g["txt_off"][i + 1]]
        for bi, b in enumerate(bands):
            for typ in range(3):
                if len(txt) < b - 1:
                    ed[i, bi, typ, 0] = -1
                    continue
                ok, s_, sk = R.banded_ed(b, typ, pat, np.concatenate([txt, np.full(64, 255, dtype=np.uint8)])[:len(txt)])
                ed[i, bi, typ] = (ok, s_, np.int64(np.int32(np.uint32(sk[0]))), np.int64(np.int32(np.uint32(sk[1]))))
    np.savez_compressed(os.path.join(HERE, "ed_golden.npz"), ed=ed, bands=np.array(bands))
    print("ed_golden.npz: %d pairs" % n)


SW_SCHEMES = ((2, -1, -2, -2), (1, -3, -4, -4), (2, -3, -5, -2), (1, -2, -1, -4))     # SimpleSmithWatermanScheme(match, mismatch, deletion, insertion)


def make_sw(R):
    """sw_golden.npz: the reference's linear-gap SmithWatermanAligner (banded: bands 3/7/15/31 x 3 types; full matrix: both
    blockings x 3 types x {no min score, the pair's min score}) and its full-matrix EditDistanceAligner on every pair of
    dp_golden.npz.  These two families stripe the full matrix 16 cells wide (sw_bandlen_selector, sw/sw_inl.h:1322-1325)
    where Gotoh uses 8: LOCAL ties and the early exit are resolved per stripe, so the outputs pin that as well."""
    g = np.load(os.path.join(HERE, "dp_golden.npz"))
    n = len(g["pat_off"]) - 1
    bands = [int(b) for b in g["bands"]]
    u = lambda v: np.int64(np.int32(np.uint32(v)))
    bsw = np.zeros((n, len(SW_SCHEMES), len(bands), 3, 4), dtype=np.int64)   # ok, score, sink.x, sink.y  (ok = -1: not pinned)
    fsw = np.zeros((n, len(SW_SCHEMES), 2, 3, 2, 4), dtype=np.int64)         # [case, scheme, blocking, type, min-score variant]
    fed = np.zeros((n, 2, 3, 2, 4), dtype=np.int64)
    for i in range(n):
        pat = g["pats"][g["pat_off"][i]:g["pat_off"][i + 1]]
        txt = g["txts"][g["txt_off"][i]:g["txt_off"][i + 1]]
        for si, sw in enumerate(SW_SCHEMES):
            for bi, b in enumerate(bands):
                for typ in range(3):
                    if len(txt) < b - 1:
                        bsw[i, si, bi, typ, 0] = -1
                        continue
                    ok, s_, sk = R.banded_sw(b, typ, sw, pat, txt)
                    bsw[i, si, bi, typ] = (ok, s_, u(sk[0]), u(sk[1]))
            for blk in range(2):
                for typ in range(3):
                    for v, ms in enumerate((oracle.SCORE_MIN, int(g["min_scores"][i]))):
                        ok, s_, sk = R.full_sw(typ, blk, sw, pat, txt, ms)
                        fsw[i, si, blk, typ, v] = (ok, s_, u(sk[0]), u(sk[1]))
        for blk in range(2):
            for typ in range(3):
                for v, ms in enumerate((oracle.SCORE_MIN, max(int(g["min_scores"][i]), -(len(pat) // 4) - 1))):
                    ok, s_, sk = R.full_ed(typ, blk, pat, txt, ms)
                    fed[i, blk, typ, v] = (ok, s_, u(sk[0]), u(sk[1]))
    np.savez_compressed(os.path.join(HERE, "sw_golden.npz"), bsw=bsw, fsw=fsw, fed=fed, bands=np.array(bands),
                        schemes=np.array(SW_SCHEMES, dtype=np.int32))
    print("sw_golden.npz: %d pairs" % n)


BEST2_DISTS = (0, 8)


def make_best2(R):
    """best2_golden.npz: every pair of dp_golden.npz scored by the reference into aln::Best2Sink<int32>(distinct_dist) --
    banded (bands 3/7/15/31 x 3 types) and full matrix (both blockings x 3 types x {no, finite} min score), for two
    distinct distances.  Out-of-band text is the 255 sentinel the reference substitutes."""
    g = np.load(os.path.join(HERE, "dp_golden.npz"))
    n = len(g["pat_off"]) - 1
    bands = [int(b) for b in g["bands"]]
    S = len(g["schemes"])
    b2 = np.zeros((n, len(BEST2_DISTS), len(bands), 3, 7), dtype=np.int64)      # ok (-1: not pinned), s1, x1, y1, s2, x2, y2
    f2 = np.zeros((n, len(BEST2_DISTS), 2, 3, 2, 7), dtype=np.int64)
    for i in range(n):
        sc = oracle.Scheme(*[int(v) for v in g["schemes"][i % S]])
        pat = g["pats"][g["pat_off"][i]:g["pat_off"][i + 1]]
        txt = g["txts"][g["txt_off"][i]:g["txt_off"][i + 1]]
        q = g["quals"][g["pat_off"][i]:g["pat_off"][i + 1]] if int(g["has_quals"][i]) else None
        for di, dist in enumerate(BEST2_DISTS):
            for bi, b in enumerate(bands):
                for typ in range(3):
                    if len(txt) < b - 1:
                        b2[i, di, bi, typ, 0] = -1
                        continue
                    ok, out = R.banded_gotoh_best2(b, typ, sc, pat, txt, q, dist)
                    b2[i, di, bi, typ] = (ok,) + out
            for blk in range(2):
                for typ in range(3):
                    for v, ms in enumerate((oracle.SCORE_MIN, int(g["min_scores"][i]))):
                        ok, out = R.full_gotoh_best2(typ, blk, sc, pat, txt, q, ms, dist)
                        f2[i, di, blk, typ, v] = (ok,) + out
    np.savez_compressed(os.path.join(HERE, "best2_golden.npz"), banded=b2, full=f2, bands=np.array(bands), dists=np.array(BEST2_DISTS))
    print("best2_golden.npz: %d pairs" % n)


def make_bt(R):
    """bt_golden.npz: nvbio::hamming_backtrack (the reference's approximate-search benchmark kernel, fmindex_test.cu:739-800)
    over the index of fm_golden.npz: 300 queries of 20-28 symbols in one 2-bit PackedStream (64 symbols of padding in front:
    the reference's traversal walks into what precedes a query, see nvbio_amd.h), text substrings with 0-2 substitutions plus
    random ones, for (seed, mismatches) in ((10,1), (12,2), (8,0)): count, number of ranges, first 48 ranges each."""
    g = np.load(os.path.join(HERE, "fm_golden.npz"))
    text = g["text"]
    O = oracle.Oracle()
    hidx = R.build_index(text)
    rng = np.random.default_rng(4242)
    Q = 300
    lens = rng.integers(20, 29, Q)
    offs = np.zeros(Q + 1, dtype=np.uint32); offs[0] = 64; offs[1:] = 64 + np.cumsum(lens)
    stream = rng.integers(0, 4, int(offs[-1]) + 64, dtype=np.uint8)
    for i in range(Q):
        if i % 6 != 5:
            p0 = int(rng.integers(0, len(text) - lens[i]))
            q = text[p0:p0 + lens[i]].copy()
            for _ in range(int(rng.integers(0, 3))):
                k = int(rng.integers(0, lens[i])); q[k] = (q[k] + 1 + rng.integers(0, 3)) % 4
            stream[offs[i]:offs[i + 1]] = q
    words = O.pack2(stream)
    modes = ((10, 1), (12, 2), (8, 0))
    CAP = 48
    counts = np.zeros((len(modes), Q), dtype=np.int64); nrs = np.zeros((len(modes), Q), dtype=np.int64)
    ranges = np.zeros((len(modes), Q, CAP, 2), dtype=np.int64)
    for mi, (seed, mm) in enumerate(modes):
        for i in range(Q):
            c, n, rg = R.hamming_backtrack(hidx, words, int(offs[i]), int(lens[i]), seed, mm, cap=CAP)
            counts[mi, i], nrs[mi, i] = c, n
            ranges[mi, i, :len(rg)] = rg
    R.destroy(hidx)
    np.savez_compressed(os.path.join(HERE, "bt_golden.npz"), stream=stream, offs=offs, modes=np.array(modes), counts=counts,
                        n_ranges=nrs, ranges=ranges)
    print("bt_golden.npz: %d queries, %d..%d ranges" % (Q, nrs.min(), nrs.max()))


def deque_cases(seed=2026, n_cases=600):
    """operation sequences on a read's seed-hit deque: nvBowtie's order (pushes under the max_hits rule, then selects) and mixed ones;
    many ties (most seeds have one occurrence), a few large ranges"""
    rng = np.random.default_rng(seed)
    cases = []
    for t in range(n_cases):
        n = int(rng.integers(1, 80)); mh = int(rng.integers(1, 24)) if t % 4 else 100
        npush = int(rng.integers(1, n + 1))
        ops = np.zeros(n, dtype=np.uint32)
        if t % 3 == 0:
            ops[:] = rng.choice([0, 0, 0, 1, 2, 3], n)
        else:
            ops[npush:] = rng.choice([3, 3, 3, 3, 1, 2], n - npush)
        begins = rng.integers(0, 3_000_000_000, n).astype(np.uint32)
        sizes = rng.choice([1, 1, 1, 1, 2, 3, 5, 40, 1000], n).astype(np.uint32)
        bits = (sizes | (rng.integers(0, 1024, n).astype(np.uint32) << 20) | (rng.integers(0, 2, n).astype(np.uint32) << 30)).astype(np.uint32)
        cases.append((ops, begins, bits, mh))
    return cases


def make_deque(R):
    """the reference's own priority_deque (nvbio/basic/priority_deque.h + interval_heap.h, the container of nvBowtie's seed-hit deques)
    driven through those sequences: the heap array as it lies in memory afterwards, and the rows the select steps returned"""
    cases = deque_cases()
    off = np.zeros(len(cases) + 1, dtype=np.int64); hoff = np.zeros(len(cases) + 1, dtype=np.int64)
    ops_all, beg_all, bit_all, rows_all, heaps, mhs = [], [], [], [], [], []
    for k, (ops, begins, bits, mh) in enumerate(cases):
        heap, rows = R.hit_deque_run(ops, begins, bits, mh)
        ops_all.append(ops); beg_all.append(begins); bit_all.append(bits); rows_all.append(rows); heaps.append(heap.reshape(-1, 2)); mhs.append(mh)
        off[k + 1] = off[k] + len(ops); hoff[k + 1] = hoff[k] + len(heap)
    np.savez_compressed(os.path.join(HERE, "deque_golden.npz"), ops=np.concatenate(ops_all), begins=np.concatenate(beg_all),
                        bits=np.concatenate(bit_all), rows=np.concatenate(rows_all), off=off, heaps=np.concatenate(heaps), hoff=hoff,
                        max_hits=np.array(mhs, dtype=np.uint32))
    print("deque_golden.npz: %d sequences, %d operations" % (len(cases), off[-1]))


def pack_words(sym, word_bits):
    """2-bit big-endian packing into 32- or 64-bit words (PackedStream<.., 2, true>)"""
    spw = word_bits // 2
    n = len(sym)
    words = np.zeros((n + spw - 1) // spw + 4, dtype=np.uint64)
    sh = (word_bits - 2 - 2 * (np.arange(n) % spw)).astype(np.uint64)
    np.bitwise_or.at(words, np.arange(n) // spw, sym.astype(np.uint64) << sh)
    return words.astype(np.uint32) if word_bits == 32 else words


def make_rankdict(R):
    """the reference's GENERIC rank dictionary (rank_dictionary_inl.h:206-336) in the two configurations of its own test
    (rank_test.cu:83-227): occurrence table, rank of every (i, c) incl. i = -1, rank4 of every i"""
    rng = np.random.default_rng(404)
    n = 7001
    sym = rng.integers(0, 4, n).astype(np.uint8)
    sym[3000:3300] = 2
    out = {"sym": sym}
    for wb in (32, 64):
        tw = pack_words(sym, wb)
        minus1 = (1 << wb) - 1
        idx = np.repeat(np.concatenate([np.arange(n, dtype=np.uint64), np.array([minus1], dtype=np.uint64)]), 4)
        cs = np.tile(np.arange(4, dtype=np.uint8), n + 1)
        occ, cnt, r, r4 = R.rank_generic(tw, wb, n, idx, cs)
        out["occ%d" % wb] = occ.astype(np.uint64); out["cnt%d" % wb] = cnt
        out["rank%d" % wb] = r.reshape(n + 1, 4); out["rank4_%d" % wb] = r4[:4 * n:4]
    np.savez_compressed(os.path.join(HERE, "rankdict_golden.npz"), **out)
    print("rankdict_golden.npz: %d symbols, 2 configurations" % n)


def myers_cases(seed=31):
    """(band, type, min_score, pattern, text) cases for the Myers aligner: near-matches at every offset of the window, indels, N's in the
    pattern, windows shorter / barely longer than the pattern, min_score values on both sides of the int16 truncation"""
    rng = np.random.default_rng(seed)
    cases = []
    for t in range(1500):
        band = int(rng.choice([3, 7, 15, 31])); typ = int(rng.choice([0, 2]))
        M = int(rng.integers(1, 170)); N = max(M + int(rng.integers(-2, 45)), 1)
        txt = rng.integers(0, 4, N).astype(np.uint8)
        if rng.random() < 0.85 and N >= M:
            off = int(rng.integers(0, N - M + 1)); pat = txt[off:off + M].copy()
            for _ in range(int(rng.integers(0, 5))):
                pat[rng.integers(0, M)] = rng.integers(0, 5)
            if rng.random() < 0.3 and M > 10:
                p = int(rng.integers(2, M - 2)); pat = np.concatenate([pat[:p], pat[p + 1:], rng.integers(0, 4, 1).astype(np.uint8)])
        else:
            pat = rng.integers(0, 5, M).astype(np.uint8)
        ms = int(rng.choice([-(1 << 30), -32768, -3, -12, -100, 0, -40000, -65536 - 7]))
        cases.append((band, typ, ms, pat, txt))
    return cases


def make_myers(R):
    cases = myers_cases()
    po = np.zeros(len(cases) + 1, dtype=np.uint32); to = np.zeros(len(cases) + 1, dtype=np.uint32)
    out = np.zeros((len(cases), 4), dtype=np.int64)
    for k, (band, typ, ms, pat, txt) in enumerate(cases):
        ok, sc, sk = R.banded_myers(band, typ, pat, txt, ms)
        out[k] = (ok, sc, sk[0], sk[1]); po[k + 1] = po[k] + len(pat); to[k + 1] = to[k] + len(txt)
    np.savez_compressed(os.path.join(HERE, "myers_golden.npz"), band=np.array([c[0] for c in cases], dtype=np.uint32),
                        typ=np.array([c[1] for c in cases], dtype=np.int32), min_score=np.array([c[2] for c in cases], dtype=np.int64),
                        pats=np.concatenate([c[3] for c in cases]), txts=np.concatenate([c[4] for c in cases]), pat_off=po, txt_off=to, out=out)
    print("myers_golden.npz: %d cases, %d report a score" % (len(cases), int((out[:, 1] > -(1 << 30)).sum())))


STAGED_SCHEMES = [(2, 2, 6, -8, -3, -8, -3), (0, 2, 6, -8, -3, -8, -3), (1, 3, 3, -5, -2, -6, -3), (0, 250, 250, -300, -150, -280, -170)]


def staged_cases(seed=77):
    """(band, type, scheme index, min_score, pattern, quals or None, text) for the staged scheduler's windowed scoring: patterns of
    1..300 rows (0..9 windows), near-matches and far-off pairs, min_score on both sides of what the pair reaches, N's, qualities, a
    scheme whose scores fall below the int16 checkpoint clamp, texts shorter than pattern + band"""
    rng = np.random.default_rng(seed)
    cases = []
    for t in range(3000):
        band = int(rng.choice([3, 7, 15, 31])); typ = int(rng.integers(0, 3))
        M = int(rng.integers(1, 300)); N = max(M + int(rng.integers(-1, band + 4)), 1)
        txt = rng.integers(0, 4, N).astype(np.uint8)
        d = int(rng.integers(0, band))
        pat = np.resize(txt[min(d, N - 1):], M).copy()
        mut = rng.random(M) < (0.02 if rng.random() < 0.5 else 0.3)
        pat[mut] = rng.integers(0, 4, int(mut.sum()))
        if rng.random() < 0.3 and M > 40:                          # the second half does not match at all
            pat[M // 2:] = rng.integers(0, 4, M - M // 2)
        if rng.random() < 0.1:
            pat[rng.integers(0, M)] = 4
        si = 0 if typ == 1 else int(rng.choice([1, 1, 1, 2, 3]))
        quals = rng.integers(0, 50, M).astype(np.uint8) if rng.random() < 0.5 else None
        ms = int(rng.choice([-(1 << 30), -400, -200, -60, -20, 0, 40, 100, 250]))
        cases.append((band, typ, si, ms, pat, quals, txt))
    return cases


def make_staged(R):
    cases = staged_cases()
    po = np.zeros(len(cases) + 1, dtype=np.uint32); to = np.zeros(len(cases) + 1, dtype=np.uint32)
    out = np.zeros((len(cases), 5), dtype=np.int64)
    quals = []
    for k, (band, typ, si, ms, pat, q, txt) in enumerate(cases):
        ok, sc, sk, windows = R.banded_gotoh_staged(band, typ, oracle.Scheme(*STAGED_SCHEMES[si]), pat, txt, ms, q)
        out[k] = (ok, sc, sk[0], sk[1], windows); po[k + 1] = po[k] + len(pat); to[k + 1] = to[k] + len(txt)
        quals.append(q if q is not None else np.full(len(pat), 255, np.uint8))          # 255 = "no quality string" for this case
    np.savez_compressed(os.path.join(HERE, "staged_golden.npz"), band=np.array([c[0] for c in cases], dtype=np.uint32),
                        typ=np.array([c[1] for c in cases], dtype=np.int32), scheme=np.array([c[2] for c in cases], dtype=np.int32),
                        schemes=np.array(STAGED_SCHEMES, dtype=np.int32), min_score=np.array([c[3] for c in cases], dtype=np.int64),
                        pats=np.concatenate([c[4] for c in cases]), quals=np.concatenate(quals),
                        txts=np.concatenate([c[6] for c in cases]), pat_off=po, txt_off=to, out=out)
    print("staged_golden.npz: %d cases, %d stop early" % (len(cases), int((out[:, 0] == 0).sum())))


SWTB_SCHEMES = [(2, -3, -5, -2), (1, -1, -1, -1), (0, -1, -1, -1), (2, -2, -1, -4), (5, -4, -8, -8)]


def swtb_cases(seed=123):
    """(band, type, scheme index, pattern, text) for the traceback of the linear-gap Smith-Waterman aligner: unequal and equal deletion /
    insertion costs, the edit-distance scheme, near-matches at every diagonal of the band, indels of 1-3 symbols, far-off pairs, N's in
    the pattern, texts shorter than the pattern and barely longer"""
    rng = np.random.default_rng(seed)
    cases = []
    for t in range(2500):
        band = int(rng.choice([3, 7, 15, 31])); typ = int(rng.integers(0, 3))
        M = int(rng.integers(1, 200)); N = max(M + int(rng.integers(-1, band + 4)), band)
        txt = rng.integers(0, 4, N).astype(np.uint8)
        d = int(rng.integers(0, band))
        pat = np.resize(txt[min(d, N - 1):], M).copy()
        mut = rng.random(M) < (0.03 if rng.random() < 0.6 else 0.3)
        pat[mut] = rng.integers(0, 4, int(mut.sum()))
        if rng.random() < 0.4 and M > 12:
            p_ = int(rng.integers(3, M - 3)); g = int(rng.integers(1, 4))
            pat = (np.concatenate([pat[:p_], pat[p_ + g:], rng.integers(0, 4, g).astype(np.uint8)]) if rng.random() < 0.5
                   else np.concatenate([pat[:p_], rng.integers(0, 4, g).astype(np.uint8), pat[p_:M - g]]))
        if rng.random() < 0.1:
            pat[rng.integers(0, M)] = 4
        cases.append((band, typ, int(rng.integers(0, len(SWTB_SCHEMES))), pat, txt))
    return cases


def make_swtb(R):
    cases = swtb_cases()
    po = np.zeros(len(cases) + 1, dtype=np.uint32); to = np.zeros(len(cases) + 1, dtype=np.uint32)
    out = np.zeros((len(cases), 6), dtype=np.int64); co = np.zeros(len(cases) + 1, dtype=np.uint32)
    cigs = []
    for k, (band, typ, si, pat, txt) in enumerate(cases):
        r, sc, src, snk, ops, clips = R.banded_sw_traceback(band, typ, SWTB_SCHEMES[si], pat, txt)
        cig = oracle.cigar_from_ops(ops, clips[0], clips[1]) if r else np.zeros(0, np.uint16)
        out[k] = (1 if r else 0, sc, src[0], src[1], snk[0], snk[1])
        cigs.append(cig); co[k + 1] = co[k] + len(cig)
        po[k + 1] = po[k] + len(pat); to[k + 1] = to[k] + len(txt)
    np.savez_compressed(os.path.join(HERE, "swtb_golden.npz"), band=np.array([c[0] for c in cases], dtype=np.uint32),
                        typ=np.array([c[1] for c in cases], dtype=np.int32), scheme=np.array([c[2] for c in cases], dtype=np.int32),
                        schemes=np.array(SWTB_SCHEMES, dtype=np.int32), pats=np.concatenate([c[3] for c in cases]),
                        txts=np.concatenate([c[4] for c in cases]), pat_off=po, txt_off=to, out=out, cig_off=co,
                        cigars=np.concatenate(cigs).astype(np.uint16))
    print("swtb_golden.npz: %d cases, %d traced, %d cigar elements" % (len(cases), int(out[:, 0].sum()), int(co[-1])))


def fswtb_cases(seed=321):
    """(type, scheme index, min_score, pattern, text) for the FULL-matrix traceback of the linear-gap Smith-Waterman aligner: patterns up
    to 200 symbols (past one 64-column checkpoint and several 16-column stripes) somewhere inside texts up to 400, substitutions, indels
    of 1-4 symbols, unrelated pairs, a 2-letter alphabet (ties everywhere), N's, texts shorter than the pattern, min_score limits that
    stop the stripe sweep early"""
    rng = np.random.default_rng(seed)
    cases = []
    for t in range(2000):
        typ = int(rng.integers(0, 3))
        M = int(rng.integers(1, 200)); N = int(rng.integers(M, 400)) if rng.random() < 0.9 else int(rng.integers(1, M + 1))
        A = 2 if rng.random() < 0.2 else 4
        txt = rng.integers(0, A, N).astype(np.uint8)
        if N >= M and rng.random() < 0.8:
            s0 = int(rng.integers(0, N - M + 1))
            pat = txt[s0:s0 + M].copy()
            mut = rng.random(M) < (0.03 if rng.random() < 0.6 else 0.25)
            pat[mut] = rng.integers(0, A, int(mut.sum()))
            if rng.random() < 0.5 and M > 12:
                p_ = int(rng.integers(3, M - 3)); g = int(rng.integers(1, 5))
                pat = (np.concatenate([pat[:p_], pat[p_ + g:], rng.integers(0, A, g).astype(np.uint8)]) if rng.random() < 0.5
                       else np.concatenate([pat[:p_], rng.integers(0, A, g).astype(np.uint8), pat[p_:M - g]]))
        else:
            pat = rng.integers(0, A, M).astype(np.uint8)
        if rng.random() < 0.1:
            pat[rng.integers(0, M)] = 4
        ms = oracle.SCORE_MIN if rng.random() < 0.7 else -int(rng.integers(0, 60))
        cases.append((typ, int(rng.integers(0, len(SWTB_SCHEMES))), ms, pat, txt))
    return cases


def make_fswtb(R):
    cases = fswtb_cases()
    po = np.zeros(len(cases) + 1, dtype=np.uint32); to = np.zeros(len(cases) + 1, dtype=np.uint32)
    out = np.zeros((len(cases), 6), dtype=np.int64); co = np.zeros(len(cases) + 1, dtype=np.uint32)
    cigs = []
    for k, (typ, si, ms, pat, txt) in enumerate(cases):
        r, sc, src, snk, ops, clips = R.full_sw_traceback(typ, SWTB_SCHEMES[si], pat, txt, ms)
        cig = oracle.cigar_from_ops(ops, clips[0], clips[1]) if r else np.zeros(0, np.uint16)
        out[k] = (1 if r else 0, sc, src[0], src[1], snk[0], snk[1])
        cigs.append(cig); co[k + 1] = co[k] + len(cig)
        po[k + 1] = po[k] + len(pat); to[k + 1] = to[k] + len(txt)
    np.savez_compressed(os.path.join(HERE, "fswtb_golden.npz"), typ=np.array([c[0] for c in cases], dtype=np.int32),
                        scheme=np.array([c[1] for c in cases], dtype=np.int32), min_score=np.array([c[2] for c in cases], dtype=np.int64),
                        schemes=np.array(SWTB_SCHEMES, dtype=np.int32), pats=np.concatenate([c[3] for c in cases]),
                        txts=np.concatenate([c[4] for c in cases]), pat_off=po, txt_off=to, out=out, cig_off=co,
                        cigars=np.concatenate(cigs).astype(np.uint16))
    print("fswtb_golden.npz: %d cases, %d traced, %d cigar elements" % (len(cases), int(out[:, 0].sum()), int(co[-1])))


if __name__ == "__main__":
    if len(sys.argv) > 1:                                            # python make_golden.py fswtb swtb ...: only these
        R = oracle.Reference()
        for name in sys.argv[1:]:
            globals()["make_" + name](R)
        sys.exit(0)
    if not oracle.Reference.available():
        oracle.build()
    R = oracle.Reference()
    make_fm(R)
    make_dp(R)
    make_tb(R)
    make_ed(R)
    make_ftb(R)
    make_sw(R)
    make_best2(R)
    make_bt(R)
    make_deque(R)
    make_rankdict(R)
    make_myers(R)
    make_staged(R)
    make_swtb(R)
    make_fswtb(R)
