"""GPU parity: the on-GPU index builder (nvbio_fm_index_build) must produce exactly the index
the reference builds on the host (same BWT, primary, occ, L2, SSA), here checked against the
oracle's builder, which is itself pinned to the reference's sais-built index."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _texts():
    rng = np.random.default_rng(9)
    yield "random-1M", rng.integers(0, 4, 1 << 20, dtype=np.uint8)
    yield "random-100003", rng.integers(0, 4, 100003, dtype=np.uint8)
    for n in (1, 2, 15, 16, 17, 31, 32, 33, 63, 64, 65, 127, 128, 129, 1000):
        yield "random-%d" % n, rng.integers(0, 4, n, dtype=np.uint8)
    yield "all-A-1000", np.zeros(1000, dtype=np.uint8)
    yield "all-T-257", np.full(257, 3, dtype=np.uint8)
    yield "period-6", np.tile(np.array([0, 1, 2, 3, 3, 2], dtype=np.uint8), 700)[:4097]
    t = rng.integers(0, 4, 50000, dtype=np.uint8)
    t[20000:30000] = t[5000:15000]                      # a 10 kbp exact repeat: LCP 10,000
    yield "long-repeat", t
    t = rng.integers(0, 4, 3000, dtype=np.uint8)
    t[-40:] = 0                                          # A-run at the very end: padded-key ties
    yield "A-tail", t


def _check(amd, orc, name, text, max_lcp=0, bucket_symbols=None):
    want = orc.build_index(text)
    fmi = amd.FMIndex.build(orc.pack2(text), len(text), kmer_len=0, max_lcp=max_lcp, bucket_symbols=bucket_symbols)
    v = fmi.view()
    assert v.length == want.n, name
    assert v.primary == want.primary, name
    assert [v.L2[i] for i in range(5)] == list(want.L2), name
    b, s = fmi.arrays()
    assert np.array_equal(amd.u32(b), want.bwt_occ), name
    assert np.array_equal(amd.u32(s), want.ssa), name
    fmi.close()


def test_build_matches_oracle(amd, orc):
    for name, text in _texts():
        _check(amd, orc, name, text, max_lcp=1 << 15)


def test_build_bucketed_path(amd, orc):
    """force the 4^b-bucket path the 3 Gbp build takes (b = 2 there)"""
    rng = np.random.default_rng(10)
    text = rng.integers(0, 4, 300000, dtype=np.uint8)
    text[1000:1200] = 0
    for b in (1, 2, 4):                                  # nvbio_fm_build_options::bucket_symbols
        _check(amd, orc, "bucket-%d" % b, text, bucket_symbols=b)
        _check(amd, orc, "bucket-tiny-%d" % b, text[:70], bucket_symbols=b)


def test_build_rejects_long_repeats_when_asked(amd, orc):
    text = np.zeros(5000, dtype=np.uint8)
    with pytest.raises(amd.NvbioError):
        amd.FMIndex.build(orc.pack2(text), len(text), max_lcp=64)


def test_built_index_answers_queries(amd, orc):
    rng = np.random.default_rng(12)
    n = 1 << 20
    text = rng.integers(0, 4, n, dtype=np.uint8)
    fmi = amd.FMIndex.build(orc.pack2(text), n, kmer_len=9)
    Q, L = 50000, 22
    starts = rng.integers(0, n - L, Q)
    syms = np.concatenate([text[s:s + L] for s in starts])
    qs = amd.PackedStringSet(orc.pack2(syms), 2, Q, fixed_len=L)
    ranges = amd.u32(fmi.match(qs))
    assert (ranges[:, 0] <= ranges[:, 1]).all()
    pos = amd.u32(fmi.locate(ranges[:, 0].copy()))
    for k in range(0, Q, 97):
        assert np.array_equal(text[pos[k]:pos[k] + L], syms[k * L:(k + 1) * L])
    fmi.close()


@pytest.mark.parametrize("sa_int", [1, 2, 4, 8, 32, 64])
def test_denser_and_sparser_sa_sampling_gives_the_same_positions(amd, orc, sa_int):
    """the handle may sample the SA at any power of two (1 = the full suffix array, sized for
    288 GB of HBM): locate() and the filter return the positions of the reference's K = 16 index"""
    rng = np.random.default_rng(13)
    n = 200003
    text = rng.integers(0, 4, n, dtype=np.uint8)
    text[777:1200] = 2
    hidx = orc.build_index(text)
    fmi = amd.FMIndex.build(orc.pack2(text), n, kmer_len=6, sa_int=sa_int)
    v = fmi.view()
    assert v.sa_int == sa_int and v.ssa_words == n // sa_int + 1
    b, s = fmi.arrays()
    assert np.array_equal(amd.u32(b), hidx.bwt_occ)
    want_ssa = np.concatenate([[0xFFFFFFFF], hidx.sa[sa_int::sa_int]]).astype(np.uint32)
    assert np.array_equal(amd.u32(s), want_ssa)
    rows = np.concatenate([np.arange(0, 3000), rng.integers(0, n + 1, 50000), [hidx.primary, n]]).astype(np.uint32)
    assert np.array_equal(amd.u32(fmi.locate(rows)), orc.locate_batch(hidx, rows))
    jt = fmi.locate_ssa_iterator(rows)
    assert np.array_equal(amd.u32(fmi.lookup_ssa_iterator(jt)), orc.locate_batch(hidx, rows))
    from util import make_queries
    syms, offs = make_queries(rng, text, 5000, 4, 30, hit_every=2, n_count=20)
    total, ranges, slots = orc.filter_rank(hidx, syms, offs)
    flt = amd.FMIndexFilter()
    assert flt.rank(fmi, amd.PackedStringSet(syms, 8, 5000, offsets=offs, ranges=True)) == total
    assert np.array_equal(amd.u32(flt.locate(0, total)), orc.filter_locate(hidx, ranges, slots, 0, total))
    fmi.close()


@pytest.mark.parametrize("k", [0, 6, 9])
def test_sa_isa_verification_shortcut_is_exact(amd, orc, k):
    """verify mode (full SA + ISA + text copy): a search whose range has collapsed to one row is
    finished by comparing the pattern with the text; ranges must stay bit-identical to the
    reference's, for hits, for misses at every remaining position, for N's, for patterns that run
    off the start of the text, in both scan directions and with the complement flag"""
    from util import make_queries
    rng = np.random.default_rng(31 + k)
    n = 150001
    text = rng.integers(0, 4, n, dtype=np.uint8)
    text[5000:5600] = np.tile(np.array([1, 1, 2, 0, 3], dtype=np.uint8), 120)      # repeats: wide ranges that collapse late
    hidx = orc.build_index(text)
    fmi = amd.FMIndex.build(orc.pack2(text), n, kmer_len=k, sa_int=1, verify=True)
    Q = 30000
    syms, offs = make_queries(rng, text, Q, 2, 40, hit_every=1)
    # break a third of the hits at a random position (a miss somewhere in the middle)
    for q in range(0, Q, 3):
        pos = int(rng.integers(offs[q], offs[q + 1]))
        syms[pos] = (syms[pos] + 1 + rng.integers(0, 3)) % 4
    syms[rng.integers(0, len(syms), 300)] = 4
    # patterns that extend past the start of the text: X + text[0:m]
    for q in range(1, 400, 4):
        L = offs[q + 1] - offs[q]
        if L > 6:
            m = int(rng.integers(3, L - 2))
            syms[offs[q + 1] - m:offs[q + 1]] = text[:m]
    # ... and past its end (forward scans): text[n-m:] + X
    for q in range(2, 400, 4):
        L = offs[q + 1] - offs[q]
        if L > 6:
            m = int(rng.integers(3, L - 2))
            syms[offs[q]:offs[q] + m] = text[n - m:]
    comp = np.where(syms < 4, 3 - syms, syms).astype(np.uint8)
    for bits, packed in ((8, syms), (4, orc.pack4(syms))):
        qs = amd.PackedStringSet(packed, bits, Q, offsets=offs, ranges=True)
        want = orc.match_batch(hidx, syms, offs)
        assert np.array_equal(amd.u32(fmi.match(qs)), want)
        assert np.array_equal(amd.u32(fmi.match(qs, amd.FM_NO_VERIFY)), want)
        assert np.array_equal(amd.u32(fmi.match(qs, amd.FM_SCAN_FORWARD)), orc.match_batch(hidx, syms, offs, reverse=True))
        assert np.array_equal(amd.u32(fmi.match(qs, amd.FM_COMPLEMENT)), orc.match_batch(hidx, comp, offs))
        assert np.array_equal(amd.u32(fmi.match(qs, amd.FM_SCAN_FORWARD | amd.FM_COMPLEMENT)),
                              orc.match_batch(hidx, comp, offs, reverse=True))
    rows = rng.integers(0, n + 1, 20000).astype(np.uint32)
    assert np.array_equal(amd.u32(fmi.locate(rows)), orc.locate_batch(hidx, rows))
    fmi.close()
    with pytest.raises(amd.NvbioError):
        amd.FMIndex.build(orc.pack2(text), n, sa_int=16, verify=True)


@pytest.mark.parametrize("k,table_flags", [(0, 0), (6, 0), (9, 0), (9, 2), (9, 4), (7, 1)])
def test_match_direct_gives_the_same_hits(amd, orc, k, table_flags):
    """nvbio_fm_match_direct: searches that collapse to one SA row finish on the text and report the position.
    Checked against the reference algorithm: range sizes, and the hits of the filter expansion (value and
    order), for hits, misses at every position, N's, patterns running off either end of the text, both scan
    directions, complement; every direct position equals locate() of the reference's final row."""
    from util import make_queries
    rng = np.random.default_rng(77 + k)
    n = 150001
    text = rng.integers(0, 4, n, dtype=np.uint8)
    text[5000:5600] = np.tile(np.array([1, 1, 2, 0, 3], dtype=np.uint8), 120)
    hidx = orc.build_index(text)
    text[9000:9200] = text[20000:20200]                      # 2 and 4 copies: the direct table's groups
    for c in range(3):
        text[40000 + 300 * c:40250 + 300 * c] = text[30000:30250]
    hidx = orc.build_index(text)
    # table_flags: every form of the direct table (format 2 with groups, without context, without groups, plain table only)
    fmi = amd.FMIndex.build(orc.pack2(text), n, kmer_len=k, sa_int=1, table_flags=table_flags)
    assert fmi.supports_direct()
    Q = 30000
    syms, offs = make_queries(rng, text, Q, 2, 40, hit_every=1)
    for q in range(0, Q, 3):
        pos = int(rng.integers(offs[q], offs[q + 1]))
        syms[pos] = (syms[pos] + 1 + rng.integers(0, 3)) % 4
    syms[rng.integers(0, len(syms), 300)] = 4
    for q in range(1, 400, 4):
        L = offs[q + 1] - offs[q]
        if L > 6:
            m = int(rng.integers(3, L - 2))
            syms[offs[q + 1] - m:offs[q + 1]] = text[:m]
    for q in range(2, 400, 4):
        L = offs[q + 1] - offs[q]
        if L > 6:
            m = int(rng.integers(3, L - 2))
            syms[offs[q]:offs[q] + m] = text[n - m:]
    comp = np.where(syms < 4, 3 - syms, syms).astype(np.uint8)
    n_direct = 0
    for flags, src, rev in ((0, syms, False), (amd.FM_SCAN_FORWARD, syms, True), (amd.FM_COMPLEMENT, comp, False),
                            (amd.FM_SCAN_FORWARD | amd.FM_COMPLEMENT, comp, True)):
        qs = amd.PackedStringSet(orc.pack4(syms), 4, Q, offsets=offs, ranges=True)
        want = orc.match_batch(hidx, src, offs, reverse=rev).astype(np.int64)
        ranges, direct = fmi.match_direct(qs, flags)
        got, d = amd.u32(ranges).astype(np.int64), direct.cpu().numpy().astype(bool)
        wsize = np.where(want[:, 1] >= want[:, 0], want[:, 1] + 1 - want[:, 0], 0)
        gsize = np.where(got[:, 1] >= got[:, 0], got[:, 1] + 1 - got[:, 0], 0)
        assert np.array_equal(wsize, gsize)
        assert np.array_equal(got[~d & (wsize > 0)], want[~d & (wsize > 0)])        # untouched queries keep their SA range
        assert (wsize[d] == 1).all()
        assert np.array_equal(got[d, 0], hidx.sa[want[d, 0]].astype(np.int64))      # position == SA[final row]
        n_direct += int(d.sum())
        # the filter expansion gives the same hits in the same order as the plain path
        flt = amd.FMIndexFilter()
        total = flt.rank_ranges(fmi, ranges, direct)
        plain = amd.FMIndexFilter()
        assert plain.rank_ranges(fmi, fmi.match(qs, flags)) == total == int(wsize.sum())
        assert np.array_equal(amd.u32(flt.locate(0, total)), amd.u32(plain.locate(0, total)))
        b, e = total // 3, total // 3 + 1001
        assert np.array_equal(amd.u32(flt.locate(b, e)), amd.u32(plain.locate(0, total))[b:e])
    assert n_direct > 5000                                                           # the direct route was really taken
    fmi.close()
    # an adopted index (no text) cannot serve it
    adopted = amd.FMIndex.from_arrays(hidx.n, hidx.primary, hidx.L2, hidx.bwt_occ, hidx.ssa)
    assert not adopted.supports_direct()
    with pytest.raises(amd.NvbioError):
        adopted.match_direct(amd.PackedStringSet(orc.pack4(syms), 4, Q, offsets=offs, ranges=True))
    adopted.close()


def test_reference_index_files_roundtrip(amd, orc, tmp_path):
    """.bwt / .sa in the reference's on-disk format (fmindex_impl.cu:111-252; writers nvBWT.cu:303-342):
    files written independently from the oracle's index load into the oracle's arrays, and an index
    built on the GPU survives save -> load"""
    rng = np.random.default_rng(41)
    for n in (100003, 64, 65, 4097):
        text = rng.integers(0, 4, n, dtype=np.uint8)
        hidx = orc.build_index(text)
        # independent writer, straight from the format description
        words = (n + 15) // 16
        bwt_words = hidx.bwt_occ.reshape(-1, 8)[:, :4].reshape(-1)[:words]
        bwt_file, sa_file = str(tmp_path / ("g%d.bwt" % n)), str(tmp_path / ("g%d.sa" % n))
        with open(bwt_file, "wb") as f:
            np.array([hidx.primary, hidx.L2[1], hidx.L2[2], hidx.L2[3], hidx.L2[4]], dtype=np.uint32).tofile(f)
            bwt_words.astype(np.uint32).tofile(f)
        with open(sa_file, "wb") as f:
            np.array([hidx.primary, hidx.L2[1], hidx.L2[2], hidx.L2[3], hidx.L2[4], 16, n], dtype=np.uint32).tofile(f)
            hidx.ssa[1:].astype(np.uint32).tofile(f)
        fmi = amd.FMIndex.load(bwt_file, sa_file, kmer_len=4)
        v = fmi.view()
        assert (v.length, v.primary, v.sa_int) == (n, hidx.primary, 16)
        assert [v.L2[i] for i in range(5)] == list(hidx.L2)
        b, s = fmi.arrays()
        assert np.array_equal(amd.u32(b), hidx.bwt_occ) and np.array_equal(amd.u32(s), hidx.ssa)
        rows = rng.integers(0, n + 1, 2000).astype(np.uint32)
        assert np.array_equal(amd.u32(fmi.locate(rows)), orc.locate_batch(hidx, rows))
        fmi.close()
        # GPU build -> save -> byte-identical files -> load
        built = amd.FMIndex.build(orc.pack2(text), n, sa_int=16)
        b2, s2 = str(tmp_path / "b.bwt"), str(tmp_path / "b.sa")
        built.save(b2, s2)
        assert open(b2, "rb").read() == open(bwt_file, "rb").read()
        assert open(s2, "rb").read() == open(sa_file, "rb").read()
        built.close()
        # a match-only index (no .sa)
        m = amd.FMIndex.load(bwt_file)
        assert m.view().ssa_words == 0
        with pytest.raises(amd.NvbioError):
            m.locate(rows)
        m.close()
    with pytest.raises(amd.NvbioError):
        amd.FMIndex.load(str(tmp_path / "missing.bwt"))
