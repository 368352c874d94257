"""GPU parity of the generic rank dictionary (nvbio_rank_dictionary_*: plain 32- / 64-bit words, separate occurrence table, any K, 32- or
64-bit indices) against the reference's own outputs (tests/golden/rankdict_golden.npz, the two configurations of nvbio-test/rank_test.cu),
against the oracle on other (word, K, index) combinations, and -- the reason 64-bit indices exist -- on a text beyond 2^32 symbols
through size-independent properties."""
import importlib.util
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _mg():
    spec = importlib.util.spec_from_file_location("mg", os.path.join(os.path.dirname(__file__), "golden", "make_golden.py"))
    mg = importlib.util.module_from_spec(spec); spec.loader.exec_module(mg)
    return mg


def _dev(a):
    import torch
    return torch.from_numpy(a.view(np.int32) if a.dtype == np.uint32 else a.view(np.int64)).cuda()


def test_generic_rank_golden(amd, rankdict_golden):
    import torch
    g = rankdict_golden
    sym = g["sym"]; n = len(sym)
    mg = _mg()
    for wb, K in ((32, 64), (64, 128)):
        tw = mg.pack_words(sym, wb)
        rd = amd.RankDictionary(_dev(tw), n, K, wb)
        nb = (n + K - 1) // K
        occ = rd.occ.cpu().numpy().view(np.uint32 if wb == 32 else np.uint64)[:4 * nb].reshape(nb, 4).astype(np.uint64)
        assert np.array_equal(occ, g["occ%d" % wb]) and rd.counts == [int(c) for c in g["cnt%d" % wb]]
        ity = np.uint32 if wb == 32 else np.uint64
        idx = np.repeat(np.concatenate([np.arange(n, dtype=np.uint64), np.array([(1 << wb) - 1], dtype=np.uint64)]).astype(ity), 4)
        cs = np.tile(np.arange(4, dtype=np.uint8), n + 1)
        r = rd.rank(_dev(idx), torch.from_numpy(cs).cuda()).cpu().numpy().view(ity).astype(np.uint64)
        assert np.array_equal(r.reshape(n + 1, 4), g["rank%d" % wb])
        r4 = rd.rank4(_dev(np.arange(n).astype(ity))).cpu().numpy().view(ity).astype(np.uint64)
        assert np.array_equal(r4, g["rank4_%d" % wb])


@pytest.mark.parametrize("wb,K,ib", [(32, 32, 32), (32, 128, 64), (64, 64, 32), (64, 256, 64), (32, 64, 64)])
def test_generic_rank_other_shapes_equal_the_oracle(amd, orc, wb, K, ib):
    import torch
    rng = np.random.default_rng(wb + K + ib)
    n = 100_003
    sym = rng.integers(0, 4, n).astype(np.uint8)
    sym[5000:9000] = 1
    tw = _mg().pack_words(sym, wb)
    rd = amd.RankDictionary(_dev(tw), n, K, ib)
    ity = np.uint32 if ib == 32 else np.uint64
    q = np.concatenate([rng.integers(0, n, 20000).astype(np.uint64), np.array([0, n - 1, K - 1, K, (1 << ib) - 1], dtype=np.uint64)]).astype(ity)
    cs = rng.integers(0, 4, len(q)).astype(np.uint8)
    got = rd.rank(_dev(q), torch.from_numpy(cs).cuda()).cpu().numpy().view(ity).astype(np.uint64)
    occ, cnt, want = orc.rank_generic(tw, wb, n, K, ib, q.astype(np.uint64), cs)
    assert np.array_equal(got, want) and rd.counts == [int(c) for c in cnt]
    nb = (n + K - 1) // K
    assert np.array_equal(rd.occ.cpu().numpy().view(ity)[:4 * nb].reshape(nb, 4), occ)


def test_rank_beyond_2_to_32_symbols(amd):
    """5 G symbols (1.25 GB of 64-bit words), K = 128, 64-bit indices: the counts partition the positions, ranks grow by exactly the
    symbol at the position, and the totals add up -- at indices on both sides of 2^32"""
    import torch
    n = 5_000_000_000
    g = torch.Generator(device="cuda:0"); g.manual_seed(5)
    words = torch.randint(-2 ** 63, 2 ** 63 - 1, ((n + 31) // 32 + 8,), dtype=torch.int64, device="cuda:0", generator=g)
    rd = amd.RankDictionary(words, n, 128, 64)
    assert sum(rd.counts) == n and all(abs(c - n / 4) < 1e6 for c in rd.counts)
    idx = torch.cat([torch.randint(0, n - 1, (200_000,), device="cuda:0", generator=g, dtype=torch.int64),
                     torch.tensor([0, 2 ** 32 - 2, 2 ** 32 - 1, 2 ** 32, 2 ** 32 + 1, n - 2], device="cuda:0")])
    r4a, r4b = rd.rank4(idx), rd.rank4(idx + 1)
    assert torch.equal(r4a.sum(dim=1), idx + 1)                                    # every position holds exactly one symbol
    w = words[(idx + 1) >> 5]
    symb = (w >> (62 - 2 * ((idx + 1) & 31))) & 3                                  # the symbol at idx + 1
    step = r4b - r4a
    assert torch.equal(step.sum(dim=1), torch.ones_like(idx))
    assert torch.equal(step.gather(1, symb[:, None]).view(-1), torch.ones_like(idx))
    last = rd.rank4(torch.tensor([n - 1], device="cuda:0"))
    assert [int(v) for v in last[0]] == rd.counts
    r1 = rd.rank(idx, symb.to(torch.uint8))
    assert torch.equal(r1, r4a.gather(1, symb[:, None]).view(-1))
