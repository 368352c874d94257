"""Live fuzz of the oracle against the reference's own host code (oracle/_ref, built from
/root/reference by oracle/Makefile).  Skipped where the reference is not present (GPU box)."""
import numpy as np

import oracle

SCHEMES = [oracle.Scheme.simple(2, -1, -1, -1), oracle.Scheme.simple(0, -5, -8, -3),
           oracle.Scheme.simple(2, -1, -2, -1), oracle.Scheme(2, 2, 6, -8, -3, -8, -3),
           oracle.Scheme(0, 2, 6, -8, -3, -8, -3), oracle.Scheme(1, 3, 3, -15, -4, -11, -2)]


def test_fm_index_fuzz(orc, ref):
    rng = np.random.default_rng(7)
    for n in (1, 2, 15, 16, 17, 63, 64, 65, 1000, 4097, 50000):
        text = rng.integers(0, 4, n, dtype=np.uint8)
        if n == 1000:
            text[:] = 0
        if n == 4097:
            text = np.tile(np.array([0, 1, 2, 3, 3, 2], dtype=np.uint8), 700)[:n]
        ri, oi = ref.build_index(text), orc.build_index(text)
        assert np.array_equal(ri.sa[1:], oi.sa[1:])
        assert ri.primary == oi.primary and np.array_equal(ri.L2, oi.L2) and np.array_equal(ri.ssa, oi.ssa)
        ks = list(range(-1, min(n, 200) + 1)) + [int(x) for x in rng.integers(0, n + 1, 100)]
        for k in ks:
            for c in range(4):
                assert ref.rank(ri, k, c) == orc.rank(oi, k, c)
        for _ in range(200):
            l = int(rng.integers(-1, n + 1)); r = int(rng.integers(max(l, 0), n + 1)); c = int(rng.integers(0, 4))
            assert np.array_equal(ref.rank2(ri, l, r, c), orc.rank2(oi, l, r, c))
        Q = 300
        lens = rng.integers(1, 30, Q)
        offs = np.zeros(Q + 1, dtype=np.uint32); offs[1:] = np.cumsum(lens)
        syms = rng.integers(0, 4, int(offs[-1]), dtype=np.uint8)
        for q in range(0, Q, 2):
            if lens[q] <= n:
                p = int(rng.integers(0, n - lens[q] + 1)); syms[offs[q]:offs[q + 1]] = text[p:p + lens[q]]
        syms[rng.integers(0, len(syms), 10)] = 4
        for rev in (False, True):
            assert np.array_equal(ref.match_batch(ri, syms, offs, rev), orc.match_batch(oi, syms, offs, rev))
        rows = rng.integers(0, n + 1, 200).astype(np.uint32)
        assert np.array_equal(ref.locate_batch(ri, rows), orc.locate_batch(oi, rows))
        ref.destroy(ri)


def test_gotoh_fuzz(orc, ref):
    rng = np.random.default_rng(11)
    for it in range(600):
        band = [3, 7, 15, 31][it % 4]
        M = int(rng.integers(1, 160))
        N = int(rng.integers(max(M, band - 1), M + band + 10))
        if it % 7 == 0:
            N = max(band - 1, int(rng.integers(max(1, M - 3), M + 2)))
        txt = rng.integers(0, 4, N, dtype=np.uint8)
        st = int(rng.integers(0, max(1, N - M + 1)))
        pat = txt[st:st + M].copy()
        if len(pat) < M:
            pat = np.concatenate([pat, rng.integers(0, 4, M - len(pat), dtype=np.uint8)])
        mut = rng.random(M) < 0.08
        pat[mut] = rng.integers(0, 5, int(mut.sum()))
        quals = rng.integers(0, 60, M, dtype=np.uint8) if it % 2 else None
        sc = SCHEMES[it % len(SCHEMES)]
        for typ in range(3):
            assert ref.banded_gotoh(band, typ, sc, pat, txt, quals) == orc.banded_gotoh(band, typ, sc, pat, txt, quals)
            if quals is None:   # edit distance == Gotoh(0, -1, -1, -1) (ed/ed_banded_inl.h:37-69 -> sw/sw_banded_inl.h)
                assert ref.banded_ed(band, typ, pat, txt) == orc.banded_gotoh(band, typ, oracle.Scheme(*oracle.ED_SCHEME), pat, txt)
            # traceback: Alignment {score, source, sink}, the op string and the clips (banded_inl.h:354-417)
            r, rs, rsrc, rsnk, rops, rclips = ref.banded_gotoh_traceback(band, typ, sc, pat, txt, quals)
            ok, s, src, snk, cig, ops = orc.banded_gotoh_traceback(band, typ, sc, pat, txt, quals)
            assert (ok, s, src, snk) == (1 if r == 2 else 0, rs, rsrc, rsnk), (it, typ)
            assert np.array_equal(ops, rops), (it, typ)
            assert np.array_equal(cig, oracle.cigar_from_ops(rops, *rclips) if r == 2 else np.zeros(0, dtype=np.uint16)), (it, typ)
            ms = oracle.SCORE_MIN if it % 3 else int(rng.integers(-50, 200))
            # full-matrix traceback (alignment_inl.h:355-455)
            r, rs, rsrc, rsnk, rops, rclips = ref.full_gotoh_traceback(typ, sc, pat, txt, quals, ms)
            ok, s, src, snk, cig = orc.full_gotoh_traceback(typ, sc, pat, txt, quals, ms)
            assert (ok, s, src, snk) == (1 if r == 2 else 0, rs, rsrc, rsnk), (it, typ)
            assert np.array_equal(cig, oracle.cigar_from_ops(rops, *rclips) if r == 2 else np.zeros(0, dtype=np.uint16)), (it, typ)
            for blk in range(2):
                assert ref.full_gotoh(typ, blk, sc, pat, txt, quals, ms) == orc.full_gotoh(typ, blk, sc, pat, txt, quals, ms)


def test_whole_path_oracle_equals_reference_code(orc, ref):
    """the CPU composition used as bench.py's baseline: the oracle's functions and the reference's own
    host templates (on an adopted index) give the same best alignment for every read"""
    from oracle import cpu_pipeline
    rng = np.random.default_rng(3)
    G = 300000
    text = rng.integers(0, 4, G, dtype=np.uint8)
    text[1000:1500] = text[200000:200500]
    hidx = orc.build_index(text)
    ridx = ref.adopt_index(oracle.HostIndex(hidx.n, hidx.primary, hidx.L2, hidx.bwt_occ, hidx.ssa))
    R, M = 2000, 150
    starts = rng.integers(0, G - M, R)
    starts[:40] = rng.integers(1000, 1300, 40)
    reads = np.stack([text[s:s + M] for s in starts]).copy()
    reads[rng.random(reads.shape) < 0.02] = 2
    rc = rng.random(R) < 0.5
    reads[rc] = 3 - reads[rc][:, ::-1]
    a = cpu_pipeline.seed_and_extend_cpu(orc, hidx, text, G, reads)
    b = cpu_pipeline.seed_and_extend_ref(ref, orc, ridx, orc.pack2(text), G, reads)
    for x, y in zip(a[:3], b[:3]):
        assert np.array_equal(x, y)
    assert a[3] == b[3]
    ref.destroy(ridx)


def test_seed_hit_deque_fuzz(orc, ref):
    """live: the oracle's interval heap against the reference's priority_deque on fresh random operation sequences"""
    rng = np.random.default_rng(77)
    for t in range(1500):
        n = int(rng.integers(1, 120)); mh = int(rng.integers(1, 40))
        ops = rng.choice([0, 0, 0, 1, 2, 3, 3], n).astype(np.uint32)
        begins = rng.integers(0, 1 << 32, n, dtype=np.uint64).astype(np.uint32)
        sizes = rng.choice([1, 1, 1, 2, 7, 300], n).astype(np.uint32)
        bits = (sizes | (rng.integers(0, 1024, n).astype(np.uint32) << 20)).astype(np.uint32)
        h1, r1 = orc.hit_deque_run(ops, begins, bits, mh)
        h2, r2 = ref.hit_deque_run(ops, begins, bits, mh)
        assert np.array_equal(h1, h2) and np.array_equal(r1, r2), t
