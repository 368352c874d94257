"""GPU parity: FM-index match / rank / locate / filter through the C-ABI vs the oracle and the
golden vectors the reference produced.  Bit-exact (uint32 SA ranges, rows and positions)."""
import numpy as np
import pytest

import oracle
from util import make_queries

pytestmark = pytest.mark.gpu


def _golden_dev_index(amd, g, k=0):
    return amd.FMIndex.from_arrays(len(g["text"]), int(g["primary"]), g["L2"], g["bwt_occ"], g["ssa"], kmer_len=k)


@pytest.mark.parametrize("k", [0, 1, 3, 6])
def test_match_golden(amd, orc, fm_golden, k):
    g = fm_golden
    fmi = _golden_dev_index(amd, g, k)
    Q = len(g["q_offs"]) - 1
    # byte-per-symbol, concatenated ranges
    qs = amd.PackedStringSet(g["q_syms"], 8, Q, offsets=g["q_offs"], ranges=True)
    assert np.array_equal(amd.u32(fmi.match(qs)), g["ranges_bwd"])
    assert np.array_equal(amd.u32(fmi.match(qs, amd.FM_SCAN_FORWARD)), g["ranges_fwd"])
    # the same queries 4-bit packed (N = 4 survives) and, without N's, 2-bit packed
    qs4 = amd.PackedStringSet(orc.pack4(g["q_syms"]), 4, Q, offsets=g["q_offs"], ranges=True)
    assert np.array_equal(amd.u32(fmi.match(qs4)), g["ranges_bwd"])
    assert np.array_equal(amd.u32(fmi.match(qs4, amd.FM_SCAN_FORWARD)), g["ranges_fwd"])
    clean = g["q_syms"].copy()
    clean[clean > 3] = 0
    idx = oracle.HostIndex(len(g["text"]), int(g["primary"]), g["L2"], g["bwt_occ"], g["ssa"])
    want = orc.match_batch(idx, clean, g["q_offs"])
    qs2 = amd.PackedStringSet(orc.pack2(clean), 2, Q, offsets=g["q_offs"], ranges=True)
    assert np.array_equal(amd.u32(fmi.match(qs2)), want)
    # complement flag == matching the complemented string
    comp = np.where(g["q_syms"] < 4, 3 - g["q_syms"], g["q_syms"]).astype(np.uint8)
    want_c = orc.match_batch(idx, comp, g["q_offs"])
    assert np.array_equal(amd.u32(fmi.match(qs, amd.FM_COMPLEMENT)), want_c)
    fmi.close()


def test_rank_every_row_golden(amd, fm_golden):
    g = fm_golden
    fmi = _golden_dev_index(amd, g)
    n = len(g["text"])
    rows = np.repeat(np.arange(-1, n + 1, dtype=np.int64), 4).astype(np.uint32)
    syms = np.tile(np.arange(4, dtype=np.uint8), n + 2)
    got = amd.u32(fmi.rank(rows, syms)).reshape(n + 2, 4)
    assert np.array_equal(got, g["ranks"])
    got4 = amd.u32(fmi.rank4(np.arange(0, n + 1, dtype=np.uint32)))
    assert np.array_equal(got4, g["ranks4"])
    fmi.close()


def test_basic_inv_psi(amd, orc, fm_golden):
    """one LF step (fmindex_inl.h:286-309): SA[inv_psi(i)] == SA[i] - 1, and equal to the oracle for every row"""
    g = fm_golden
    fmi = _golden_dev_index(amd, g)
    n = len(g["text"])
    rows = np.arange(0, n + 1, dtype=np.uint32)
    got = amd.u32(fmi.basic_inv_psi(rows))
    idx = oracle.HostIndex(n, int(g["primary"]), g["L2"], g["bwt_occ"], g["ssa"])
    assert np.array_equal(got, orc.basic_inv_psi(idx, rows))
    sa = g["sa"].astype(np.int64); sa[0] = n
    nz = sa[rows] > 0
    assert np.array_equal(sa[got[nz]], sa[rows[nz]] - 1)
    fmi.close()


def test_locate_golden(amd, fm_golden):
    g = fm_golden
    fmi = _golden_dev_index(amd, g)
    assert np.array_equal(amd.u32(fmi.locate(g["rows"])), g["pos"])
    jt = fmi.locate_ssa_iterator(g["rows"])
    assert np.array_equal(amd.u32(jt), g["jt"])
    assert np.array_equal(amd.u32(fmi.lookup_ssa_iterator(jt)), g["pos"])
    fmi.close()


@pytest.fixture(scope="module")
def big(orc):
    rng = np.random.default_rng(42)
    n = 1 << 20
    text = rng.integers(0, 4, n, dtype=np.uint8)
    text[5000:9000] = np.tile(np.array([2, 3, 2, 2, 0], dtype=np.uint8), 800)      # repeats -> wide ranges
    return rng, text, orc.build_index(text)


@pytest.mark.parametrize("k", [0, 8, 10])
def test_match_locate_1M(amd, orc, big, k):
    rng, text, hidx = big
    fmi = amd.FMIndex.from_arrays(hidx.n, hidx.primary, hidx.L2, hidx.bwt_occ, hidx.ssa, kmer_len=k)
    Q = 100000
    syms, offs = make_queries(rng, text, Q, 1, 40, hit_every=3, n_count=200)
    want, wblocks = orc.match_batch(hidx, syms, offs, want_blocks=True)
    qs = amd.PackedStringSet(orc.pack4(syms), 4, Q, offsets=offs, ranges=True)
    assert np.array_equal(amd.u32(fmi.match(qs)), want)
    want_f = orc.match_batch(hidx, syms, offs, reverse=True)
    assert np.array_equal(amd.u32(fmi.match(qs, amd.FM_SCAN_FORWARD)), want_f)
    r, blocks = fmi.match(qs, amd.FM_NO_KMER_TABLE, want_blocks=True)
    assert np.array_equal(amd.u32(r), want)
    assert np.array_equal(amd.u32(blocks), wblocks)         # algorithmic traffic unit
    # fixed-length seeds addressed by start offset inside a read stream (nvBowtie seeds)
    L = 22
    starts = rng.integers(0, len(syms) - L, 50000).astype(np.uint32)
    qs_f = amd.PackedStringSet(orc.pack4(syms), 4, len(starts), offsets=starts, fixed_len=L)
    soffs = (np.arange(len(starts) + 1) * L).astype(np.uint32)
    ssyms = np.concatenate([syms[s:s + L] for s in starts])
    assert np.array_equal(amd.u32(fmi.match(qs_f)), orc.match_batch(hidx, ssyms, soffs))
    rc = np.where(ssyms < 4, 3 - ssyms, ssyms).astype(np.uint8)
    assert np.array_equal(amd.u32(fmi.match(qs_f, amd.FM_COMPLEMENT)), orc.match_batch(hidx, rc, soffs))
    # locate: every row of the non-empty ranges (capped), and random rows
    rows = np.concatenate([rng.integers(0, hidx.n + 1, 200000), [0, hidx.primary, hidx.n]]).astype(np.uint32)
    assert np.array_equal(amd.u32(fmi.locate(rows)), orc.locate_batch(hidx, rows))
    nz = rows > 0
    assert np.array_equal(amd.u32(fmi.locate(rows))[nz], hidx.sa[rows[nz]])        # == the true SA
    fmi.close()


def test_filter_rank_locate(amd, orc, big):
    rng, text, hidx = big
    fmi = amd.FMIndex.from_arrays(hidx.n, hidx.primary, hidx.L2, hidx.bwt_occ, hidx.ssa, kmer_len=8)
    Q = 20000
    syms, offs = make_queries(rng, text, Q, 6, 30, hit_every=2, n_count=50)
    total, ranges, slots = orc.filter_rank(hidx, syms, offs)
    flt = amd.FMIndexFilter()
    qs = amd.PackedStringSet(syms, 8, Q, offsets=offs, ranges=True)
    assert flt.rank(fmi, qs) == total == flt.n_hits()
    assert np.array_equal(amd.u32(flt.ranges()), ranges)
    assert np.array_equal(flt.slots().cpu().numpy().view(np.uint64), slots)
    for b, e in ((0, min(total, 50000)), (max(0, total - 1000), total), (total // 2, min(total, total // 2 + 7))):
        if e > b:
            assert np.array_equal(amd.u32(flt.locate(b, e)), orc.filter_locate(hidx, ranges, slots, b, e))
    fmi.close()


def test_empty_and_error_paths(amd, fm_golden):
    g = fm_golden
    fmi = _golden_dev_index(amd, g)
    qs = amd.PackedStringSet(np.zeros(4, dtype=np.uint8), 8, 0, fixed_len=4)
    assert fmi.match(qs).shape[0] == 0
    # zero-length queries: the whole index (0, n)
    qs0 = amd.PackedStringSet(np.zeros(4, dtype=np.uint8), 8, 3, fixed_len=0, stride=0)
    assert np.array_equal(amd.u32(fmi.match(qs0)), np.array([[0, len(g["text"])]] * 3, dtype=np.uint32))
    with pytest.raises(amd.NvbioError):
        amd.FMIndex.from_arrays(len(g["text"]), int(g["primary"]), g["L2"], g["bwt_occ"][:8], g["ssa"])
    with pytest.raises(amd.NvbioError):
        amd.FMIndex.from_arrays(len(g["text"]), int(g["primary"]), g["L2"], g["bwt_occ"], g["ssa"], kmer_len=18)
    fmi.close()


def test_seed_enumeration_and_diagonal_helpers(amd, orc, big):
    """seeds enumerated inside the kernel (uniform_seeds_functor) == the same seeds passed as explicit
    infixes; hit_to_diagonal / genome_infixes helpers == their definition (fmmap.cu:92-117,169-196)"""
    import torch
    rng, text, hidx = big
    fmi = amd.FMIndex.from_arrays(hidx.n, hidx.primary, hidx.L2, hidx.bwt_occ, hidx.ssa, kmer_len=8)
    R, M, L, S = 3000, 150, 22, 15
    spr = (M - L) // S + 1
    starts = rng.integers(0, hidx.n - M, R)
    reads = np.stack([text[s:s + M] for s in starts]).copy()
    reads[rng.random(reads.shape) < 0.01] = 4
    flat4 = orc.pack4(reads.reshape(-1))
    offs = (np.arange(R)[:, None] * M + np.arange(spr)[None, :] * S).reshape(-1).astype(np.uint32)
    explicit = amd.PackedStringSet(flat4, 4, R * spr, offsets=offs, fixed_len=L)
    enumerated = amd.PackedStringSet(flat4, 4, R * spr, fixed_len=L, stride=M, seeds_per_string=spr, seed_interval=S)
    for flags in (0, amd.FM_SCAN_FORWARD | amd.FM_COMPLEMENT):
        assert torch.equal(fmi.match(explicit, flags), fmi.match(enumerated, flags))
    # with per-read offsets (ragged read stream)
    roffs = (np.arange(R + 1) * M).astype(np.uint32)
    enum2 = amd.PackedStringSet(flat4, 4, R * spr, offsets=roffs, fixed_len=L, seeds_per_string=spr, seed_interval=S)
    assert torch.equal(fmi.match(explicit), fmi.match(enum2))

    hits = np.stack([rng.integers(0, 2 ** 32 - 1, 5000, dtype=np.uint64).astype(np.uint32),
                     rng.integers(0, R * spr, 5000).astype(np.uint32)], axis=1)
    hits[:50, 0] = rng.integers(0, 40, 50)                         # diagonals below zero
    ht = torch.from_numpy(hits.view(np.int32)).cuda()
    G = 3_000_000_000
    for strand in (0, 1):
        keys = amd.hits_to_diagonals(ht, spr, S, L, M, strand)
        rid = hits[:, 1].astype(np.int64) // spr
        p = (hits[:, 1].astype(np.int64) % spr) * S
        if strand:
            p = M - p - L
        want = (rid << 34) | (strand << 33) | (hits[:, 0].astype(np.int64) - p + 1024)
        assert np.array_equal(keys.cpu().numpy(), want)
        r, fl, wb, we = amd.diagonals_to_windows(keys, 31, M, G)
        g = np.maximum(hits[:, 0].astype(np.int64) - p, 0)
        wwb = np.where(g > 15, g - 15, 0)
        assert np.array_equal(amd.u32(wb).astype(np.int64), wwb)
        assert np.array_equal(amd.u32(we).astype(np.int64), np.minimum(wwb + 31 + M, G))
        assert np.array_equal(amd.u32(r).astype(np.int64), rid)
        assert (fl.cpu().numpy() == (3 if strand else 0)).all()
    # the fused forms: filter expansion straight to diagonal keys; per-read best candidate by atomic max
    for strand, flags in ((0, 0), (1, amd.FM_SCAN_FORWARD | amd.FM_COMPLEMENT)):
        flt = amd.FMIndexFilter()
        total = flt.rank(fmi, enumerated, flags)
        assert total > 0 or strand == 1                             # the reads are forward copies: few reverse-strand hits
        if total == 0:
            continue
        keys = flt.locate_diagonals(0, total, spr, S, L, M, strand)
        assert torch.equal(keys, amd.hits_to_diagonals(flt.locate(0, total), spr, S, L, M, strand))
        b, e = total // 4, total // 4 + 777
        assert torch.equal(flt.locate_diagonals(b, e, spr, S, L, M, strand), keys[b:e])
    n = 20000
    ck = torch.from_numpy(((rng.integers(0, 500, n).astype(np.int64) << 34) | (rng.integers(0, 2, n).astype(np.int64) << 33))).cuda()
    sc_ = torch.from_numpy(rng.integers(-200, 300, n).astype(np.int32)).cuda()
    sk_ = torch.from_numpy(rng.integers(0, 181, (n, 2)).astype(np.int32)).cuda()
    wb_ = torch.from_numpy(rng.integers(0, 2 ** 31 - 200, n).astype(np.int32)).cuda()
    best = amd.best_candidate_reduce(ck, sc_, sk_, wb_, torch.zeros(500, dtype=torch.int64, device="cuda:0")).cpu().numpy()
    sel = ((sc_.cpu().numpy().astype(np.int64) + (1 << 20)) << 34) | (ck.cpu().numpy() & (1 << 33)) | \
          (wb_.cpu().numpy().astype(np.int64) + sk_.cpu().numpy()[:, 0])
    want = np.zeros(500, dtype=np.int64)
    np.maximum.at(want, ck.cpu().numpy() >> 34, sel)
    assert np.array_equal(best, want)
    fmi.close()


def test_hamming_backtrack(amd, orc, fm_golden, bt_golden):
    """nvbio_fm_hamming_backtrack: reference-quirks mode against the reference's own outputs (bt_golden.npz: counts, numbers
    of ranges, ranges in delegate order), default mode against the oracle (itself checked against a brute-force Hamming scan),
    2-bit, 4-bit and byte queries; then 20,000 queries on a larger index against the oracle"""
    g, b = fm_golden, bt_golden
    fmi = _golden_dev_index(amd, g, 4)
    stream, offs = b["stream"], b["offs"].astype(np.uint32)
    hidx = oracle.HostIndex(len(g["text"]), int(g["primary"]), g["L2"], g["bwt_occ"], g["ssa"])
    Q = len(offs) - 1
    for bits, sym in ((2, orc.pack2(stream)), (4, orc.pack4(stream)), (8, stream)):
        qs = amd.PackedStringSet(sym, bits, Q, offsets=offs, ranges=True)
        for mi, (seed, mm) in enumerate(b["modes"]):
            cnt, nr, rg = fmi.hamming_backtrack(qs, int(seed), int(mm), quirks=True, max_ranges=48)
            assert np.array_equal(amd.u32(cnt).astype(np.int64), b["counts"][mi]) and np.array_equal(amd.u32(nr).astype(np.int64), b["n_ranges"][mi])
            got = amd.u32(rg).astype(np.int64)
            for i in range(Q):
                k = min(int(b["n_ranges"][mi, i]), 48)
                assert np.array_equal(got[i, :k], b["ranges"][mi, i, :k]), (bits, mi, i)
            cnt, nr, _ = fmi.hamming_backtrack(qs, int(seed), int(mm))
            want = [orc.hamming_backtrack(hidx, stream, int(offs[i]), int(offs[i + 1] - offs[i]), int(seed), int(mm))[:2] for i in range(Q)]
            assert np.array_equal(amd.u32(cnt), np.array([w[0] for w in want], dtype=np.uint32))
            assert np.array_equal(amd.u32(nr), np.array([w[1] for w in want], dtype=np.uint32))
    fmi.close()
    rng = np.random.default_rng(8)
    G = 400000
    text = rng.integers(0, 4, G, dtype=np.uint8)
    hidx = orc.build_index(text)
    fmi = amd.FMIndex.from_arrays(hidx.n, hidx.primary, hidx.L2, hidx.bwt_occ, hidx.ssa, kmer_len=8)
    Q, L = 20000, 32
    starts = rng.integers(0, G - L, Q)
    qsym = np.stack([text[s:s + L] for s in starts]).copy()
    flip = rng.random(qsym.shape) < 0.03
    qsym[flip] = (qsym[flip] + 1 + rng.integers(0, 3, int(flip.sum()))) % 4
    flat = np.concatenate([rng.integers(0, 4, 64, dtype=np.uint8), qsym.reshape(-1)])
    offs = (64 + np.arange(Q + 1) * L).astype(np.uint32)
    qs = amd.PackedStringSet(orc.pack2(flat), 2, Q, offsets=offs, ranges=True)
    for seed, mm in ((16, 1), (14, 2)):
        for quirks in (False, True):
            cnt, nr, _ = fmi.hamming_backtrack(qs, seed, mm, quirks=quirks)
            cnt, nr = amd.u32(cnt), amd.u32(nr)
            for i in range(0, Q, 9):
                c, n, _ = orc.hamming_backtrack(hidx, flat, int(offs[i]), L, seed, mm, quirks=quirks)
                assert (int(cnt[i]), int(nr[i])) == (c, n), (seed, mm, quirks, i)
        assert (cnt > 0).mean() > 0.5
    fmi.close()
