"""GPU parity of the one-call seed pass (nvbio_fm_match_seed_diagonals) against the operators it replaces --
match() of every seed (oracle, pinned on the reference), locate() of the single-row ones (the oracle's suffix array),
hit_to_diagonal, adjacent-duplicate removal -- on inputs chosen to reach every branch of the direct table:
one-occurrence entries verified on their stored left context, 2..7-occurrence groups, plain ranges that take rank
steps, seeds shorter than the table's k, seeds longer than k + 15 (context too short: text gather), occurrences at text
position 0 and at the last words of the text, N's, both scan directions, 2/4/8-bit symbols, more than 64 seeds per
read, grids capped so that waves loop over many tiles, and every NVBIO_FM_TABLE_* form of the table."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _expected(orc, hidx, reads, M, L, S, strand):
    """(keys in seed order after adjacent dedupe, {seed id: (x, y)} of multi-row seeds) from the oracle"""
    R = reads.shape[0]
    spr = (M - L) // S + 1
    seeds = np.stack([reads[:, j * S:j * S + L] for j in range(spr)], axis=1)          # [R, spr, L]
    flat = seeds.reshape(-1)
    offs = (np.arange(R * spr + 1) * L).astype(np.uint32)
    if strand == 0:
        ranges = orc.match_batch(hidx, flat, offs).astype(np.int64)
    else:
        comp = np.where(flat < 4, 3 - flat, flat).astype(np.uint8)
        ranges = orc.match_batch(hidx, comp, offs, reverse=True).astype(np.int64)
    x, y = ranges[:, 0], ranges[:, 1]
    single = x == y
    sid = np.arange(R * spr)
    rid, j = sid // spr, sid % spr
    p = j * S
    if strand:
        p = M - p - L
    pos = np.where(single, hidx.sa[np.minimum(x, hidx.n)].astype(np.int64), 0)
    pos = np.where(single & (x == 0), hidx.n, pos)                                       # row 0 is the empty suffix
    keys = (rid.astype(np.int64) << 34) | (strand << 33) | (pos + 1024 - p)
    ks = keys[single]
    keep = np.ones(len(ks), dtype=bool)
    keep[1:] = ks[1:] != ks[:-1]
    multi = x < y
    return ks[keep], {int(s): (int(a), int(b)) for s, a, b in zip(sid[multi], x[multi], y[multi])}


def _run(amd, fmi, packed, bits, R, M, L, S, strand, grid_blocks=0, extra_flags=0):
    spr = (M - L) // S + 1
    qs = amd.PackedStringSet(packed, bits, R * spr, fixed_len=L, stride=M, seeds_per_string=spr, seed_interval=S)
    flags = ((amd.FM_SCAN_FORWARD | amd.FM_COMPLEMENT) if strand else 0) | extra_flags
    b = fmi.match_seed_diagonals(qs, flags, M, strand, grid_blocks=grid_blocks)
    nk, nr = [int(v) for v in b["counts"][:2].cpu().numpy()]
    keys = b["keys"][:nk].cpu().numpy()
    rr = amd.u32(b["ranges"][:nr])
    ids = b["ids"][:nr].cpu().numpy()
    return keys, {int(i): (int(a), int(c)) for i, (a, c) in zip(ids, rr)}


def _text(rng, n):
    text = rng.integers(0, 4, n, dtype=np.uint8)
    unit = rng.integers(0, 4, 300, dtype=np.uint8)
    for c in range(12):                                            # a 12-copy family: plain ranges, rank steps, residual seeds
        text[30000 + 5000 * c:30300 + 5000 * c] = unit
    for c in range(3):                                             # a 3-copy and a 5-copy family: small and large groups
        text[100000 + 700 * c:100200 + 700 * c] = text[1000:1200]
    for c in range(5):
        text[120000 + 900 * c:120200 + 900 * c] = text[2000:2200]
    text[140000:140400] = np.tile(np.array([0, 1], dtype=np.uint8), 200)   # a tandem repeat: many occurrences one symbol apart
    return text


def _reads(rng, text, R, M):
    n = len(text)
    starts = rng.integers(0, n - M, max(R, 440))
    starts[0:8] = 0                                                # seeds at text position 0 (nothing to their left)
    starts[8:16] = np.arange(8)
    starts[16:32] = n - M - np.arange(16)                          # ... and in the last words of the text
    starts[32:200] = 30000 + 5000 * rng.integers(0, 12, 168) + rng.integers(0, 150, 168)
    starts[200:300] = 100000 + 700 * rng.integers(0, 3, 100) + rng.integers(0, 50, 100)
    starts[300:400] = 120000 + 900 * rng.integers(0, 5, 100) + rng.integers(0, 50, 100)
    starts[400:440] = 140000 + rng.integers(0, 200, 40)
    starts = np.minimum(starts[:R], n - M)
    reads = np.stack([text[s:s + M] for s in starts]).copy()
    mut = rng.random(reads.shape) < 0.01
    reads[mut] = (reads[mut] + 1 + rng.integers(0, 3, int(mut.sum()))) % 4
    reads[rng.random(reads.shape) < 0.002] = 4                     # N's
    rcm = rng.random(R) < 0.5
    reads[rcm] = np.where(reads[rcm][:, ::-1] < 4, 3 - reads[rcm][:, ::-1], 4)
    return reads


@pytest.mark.parametrize("k,table_flags", [(9, 0), (9, 2), (9, 4), (9, 6), (6, 0), (11, 0), (0, 0)])
def test_seed_pass_equals_the_operators(amd, orc, k, table_flags):
    rng = np.random.default_rng(500 + k)
    n = 300007
    text = _text(rng, n)
    hidx = orc.build_index(text)
    fmi = amd.FMIndex.build(orc.pack2(text), n, kmer_len=k, sa_int=1, table_flags=table_flags)
    R, M = 3000, 150
    reads = _reads(rng, text, R, M)
    flat = reads.reshape(-1)
    n_multi = 0
    for L, S in ((22, 15), (8, 20), (30, 12), (32, 17), (40, 25)):      # shorter than k = 9/11, within the context, beyond it, > 32 symbols
        for strand in (0, 1):
            want_keys, want_res = _expected(orc, hidx, reads, M, L, S, strand)
            for bits, packed in ((4, orc.pack4(flat)),):
                # the default kernel choice (software-pipelined where it applies), the plain kernel, the accounting one
                for extra in (0, amd.FM_NO_PIPELINE, amd.FM_COUNT_SECTORS):
                    keys, res = _run(amd, fmi, packed, bits, R, M, L, S, strand, extra_flags=extra)
                    assert np.array_equal(keys, want_keys), (k, table_flags, L, S, strand, bits, extra)
                    assert res == want_res, (k, table_flags, L, S, strand, bits, extra)
            n_multi += len(want_res)
    assert n_multi > 500                                           # the residual route was really taken
    # a grid of 64 workgroups (256 waves) over 429 tiles: every wave loops
    want_keys, want_res = _expected(orc, hidx, reads, M, 22, 15, 1)
    keys, res = _run(amd, fmi, orc.pack4(flat), 4, R, M, 22, 15, 1, grid_blocks=64)
    assert np.array_equal(keys, want_keys) and res == want_res
    fmi.close()


def test_seed_pass_symbol_widths_and_many_seeds_per_read(amd, orc):
    rng = np.random.default_rng(77)
    n = 200003
    text = _text(rng, n)
    hidx = orc.build_index(text)
    fmi = amd.FMIndex.build(orc.pack2(text), n, kmer_len=8, sa_int=1)
    # 2-bit and 8-bit symbol streams (no N's in 2 bits)
    R, M = 1500, 150
    reads = _reads(rng, text, R, M)
    clean = np.where(reads < 4, reads, 0).astype(np.uint8)
    for strand in (0, 1):
        wk, wr = _expected(orc, hidx, clean, M, 22, 15, strand)
        keys, res = _run(amd, fmi, orc.pack2(clean.reshape(-1)), 2, R, M, 22, 15, strand)
        assert np.array_equal(keys, wk) and res == wr
        wk, wr = _expected(orc, hidx, reads, M, 22, 15, strand)
        keys, res = _run(amd, fmi, reads.reshape(-1).copy(), 8, R, M, 22, 15, strand)
        assert np.array_equal(keys, wk) and res == wr
    # long reads: 99 seeds per read (more than the 64 lanes of a wave) and 13 (4 reads per tile, 12 idle lanes)
    R, M = 300, 1000
    reads = _reads(rng, text, R, M)
    for L, S in ((20, 10), (22, 80)):
        for strand in (0, 1):
            wk, wr = _expected(orc, hidx, reads, M, L, S, strand)
            keys, res = _run(amd, fmi, orc.pack4(reads.reshape(-1)), 4, R, M, L, S, strand)
            assert np.array_equal(keys, wk) and res == wr, (L, S, strand)
    # one read, one seed
    one = text[5000:5022].copy()[None, :]
    wk, wr = _expected(orc, hidx, one, 22, 22, 1, 0)
    keys, res = _run(amd, fmi, orc.pack4(one.reshape(-1)), 4, 1, 22, 22, 1, 0)
    assert np.array_equal(keys, wk) and len(wk) == 1 and res == wr
    fmi.close()


def test_seed_pass_needs_a_direct_capable_index(amd, orc):
    rng = np.random.default_rng(5)
    text = rng.integers(0, 4, 50000, dtype=np.uint8)
    fmi = amd.FMIndex.build(orc.pack2(text), len(text), kmer_len=6, sa_int=16)
    qs = amd.PackedStringSet(orc.pack4(text[:1500]), 4, 10 * 9, fixed_len=22, stride=150, seeds_per_string=9, seed_interval=15)
    with pytest.raises(amd.NvbioError):
        fmi.match_seed_diagonals(qs, 0, 150, 0)
    fmi.close()
