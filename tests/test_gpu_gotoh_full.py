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
