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


def _check(amd, orc, name, text, max_lcp=0):
    want = orc.build_index(text)
    fmi = amd.FMIndex.build(orc.pack2(text), len(text), kmer_len=0, max_lcp=max_lcp)
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
    for b in ("1", "2", "4"):
        os.environ["NVBIO_AMD_BUILD_BUCKET_SYMBOLS"] = b
        try:
            _check(amd, orc, "bucket-" + b, text)
            _check(amd, orc, "bucket-tiny-" + b, text[:70])
        finally:
            del os.environ["NVBIO_AMD_BUILD_BUCKET_SYMBOLS"]


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
