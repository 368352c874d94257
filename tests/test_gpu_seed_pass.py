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


def _single_ids(orc, hidx, reads, M, L, S, strand):
    """ids of the seeds whose search ends on exactly one row"""
    R = reads.shape[0]
    spr = (M - L) // S + 1
    flat = np.stack([reads[:, j * S:j * S + L] for j in range(spr)], axis=1).reshape(-1)
    offs = (np.arange(R * spr + 1) * L).astype(np.uint32)
    if strand == 0:
        ranges = orc.match_batch(hidx, flat, offs).astype(np.int64)
    else:
        ranges = orc.match_batch(hidx, np.where(flat < 4, 3 - flat, flat).astype(np.uint8), offs, reverse=True).astype(np.int64)
    return np.nonzero(ranges[:, 0] == ranges[:, 1])[0]


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


def _run_both(amd, fmi, packed, bits, R, M, L, S, grid_blocks=0, flags=0, inline_hits=0):
    spr = (M - L) // S + 1
    qs = amd.PackedStringSet(packed, bits, R * spr, fixed_len=L, stride=M, seeds_per_string=spr, seed_interval=S)
    b = fmi.match_seed_diagonals_both(qs, M, flags=flags, grid_blocks=grid_blocks, inline_hits=inline_hits)
    c = [int(v) for v in b["counts"][:4].cpu().numpy()]
    assert b["keys"].numel() >= 128 * -(-R // (64 // spr)) and c[0] <= b["keys"].numel()      # 128 key slots per tile of whole reads
    keys = b["keys"][:c[0]].cpu().numpy()
    n = R * spr
    res = []
    for lo, cnt in ((0, c[1]), (n, c[2])):
        rr = amd.u32(b["ranges"][lo:lo + cnt]); ids = b["ids"][lo:lo + cnt].cpu().numpy()
        res.append({int(i): (int(a), int(d)) for i, (a, d) in zip(ids, rr)})
    sectors = int(b["counts"][4:6].cpu().numpy().view(np.uint64)[0]) if flags & amd.FM_COUNT_SECTORS else None
    return keys, res[0], res[1], sectors


def _tile_major(keys_f, keys_r, spr):
    """the order of the two-strand pass: tile by tile (64 // spr whole reads), a tile's forward keys, then its reverse-strand keys"""
    rpt = 64 // spr
    tf, tr = (keys_f >> 34) // rpt, (keys_r >> 34) // rpt
    out = []
    for t in np.union1d(tf, tr):
        out.append(keys_f[tf == t]); out.append(keys_r[tr == t])
    return np.concatenate(out) if out else np.zeros(0, np.int64)


@pytest.mark.parametrize("wide", [False, True])            # 8-byte entries (one row in line) / 16-byte entries (two)
@pytest.mark.parametrize("k", [5, 9, 11, 15])
def test_two_strand_pass_equals_the_per_strand_operators(amd, orc, k, wide):
    """nvbio_fm_match_seed_diagonals_both over the canonical table (a k-mer and its reverse complement share an entry) against match() of
    every seed and of its reverse complement by the oracle: single-occurrence entries of either orientation, groups that mix both, k-mers
    with more than 8 occurrences (k = 5: every one) and repeats longer than the seed (the per-strand fallback inside the kernel, residual
    lists), occurrences at the first and last positions of the text, N's, seed lengths from k to k + 7"""
    rng = np.random.default_rng(900 + k)
    n = 300007
    text = _text(rng, n)
    # reverse-complement copies: k-mers whose two orientations BOTH occur (a group with o = 0 and o = 1 rows; a forward and a reverse hit
    # of the same seed)
    text[160000:160300] = 3 - text[5000:5300][::-1]
    text[170000:170200] = 3 - text[100000:100200][::-1]
    hidx = orc.build_index(text)
    fmi = amd.FMIndex.build(orc.pack2(text), n, kmer_len=k, sa_int=1, table_flags=amd.FM_TABLE_CANONICAL_WIDE if wide else amd.FM_TABLE_CANONICAL)
    assert fmi.canonical
    R, M = 3000, 150
    reads = _reads(rng, text, R, M)
    reads[440:500] = np.stack([text[s:s + M] for s in 5000 + rng.integers(0, 150, 60)])       # reads from the region that also occurs reversed
    flat = reads.reshape(-1)
    n_multi = n_both = 0
    for L, S in ((k, 13), (k + 3, 15), (k + 7, 20), (k + 5, 128 // 8)):
        spr = (M - L) // S + 1
        want_f, res_f = _expected(orc, hidx, reads, M, L, S, 0)
        want_r, res_r = _expected(orc, hidx, reads, M, L, S, 1)
        want = _tile_major(want_f, want_r, spr)
        for bits, packed in ((4, orc.pack4(flat)),) + (((2, orc.pack2(np.where(flat < 4, flat, 0).astype(np.uint8))),) if L == k + 3 else ()):
            if bits == 2:          # no N in a 2-bit stream: expectations for the N-free reads
                clean = np.where(reads < 4, reads, 0).astype(np.uint8)
                wf, rf = _expected(orc, hidx, clean, M, L, S, 0); wr, rr = _expected(orc, hidx, clean, M, L, S, 1)
                w2 = _tile_major(wf, wr, spr)
            else:
                w2, rf, rr = want, res_f, res_r
            for flags, gb in ((0, 0), (amd.FM_COUNT_SECTORS, 0), (0, 64)):
                keys, gf, gr, sectors = _run_both(amd, fmi, packed, bits, R, M, L, S, grid_blocks=gb, flags=flags)
                assert np.array_equal(keys, w2), (k, L, S, bits, flags, gb)
                assert gf == rf and gr == rr, (k, L, S, bits, flags, gb)
                if sectors is not None:
                    assert R * spr * 0.5 < sectors < R * spr * 60
        # inline_hits = h: a seed on 2..h rows leaves all its keys (any order inside its tile and strand), larger ranges stay residual
        for h in (2, 4):
            keys, gf, gr, _ = _run_both(amd, fmi, orc.pack4(flat), 4, R, M, L, S, inline_hits=h)
            exp = [want_f, want_r]
            rpt = 64 // spr
            for strand, res, got in ((0, res_f, gf), (1, res_r, gr)):
                # a tile's keys of one strand must fit 64 slots: where the one-row seeds plus all rows of the 2..h-row seeds would not,
                # the latter stay residual (the kernel's wave-uniform rule)
                n_single = np.bincount((_single_ids(orc, hidx, reads, M, L, S, strand) // spr) // rpt, minlength=R // rpt + 1)
                small = {sid: xy for sid, xy in res.items() if xy[1] - xy[0] + 1 <= h}
                load = n_single.copy()
                for sid, (x, y) in small.items():
                    load[(sid // spr) // rpt] += y - x + 1
                inl = {sid: xy for sid, xy in small.items() if load[(sid // spr) // rpt] <= 64}
                assert got == {sid: xy for sid, xy in res.items() if sid not in inl}, (k, L, S, h, strand)
                for sid, (x, y) in inl.items():
                    rid, j = sid // spr, sid % spr
                    p = (M - j * S - L) if strand else j * S
                    exp.append((np.int64(rid) << 34) | (strand << 33) | (hidx.sa[x:y + 1].astype(np.int64) + 1024 - p))
            assert np.array_equal(np.sort(keys), np.sort(np.concatenate(exp))), (k, L, S, h)
            # ... and the keys are still grouped tile by tile, forward strand first
            order = ((keys >> 34) // rpt) * 2 + ((keys >> 33) & 1)
            assert (np.diff(order) >= 0).all()
        n_multi += len(res_f) + len(res_r)
        n_both += len(np.intersect1d(want_f >> 34, want_r >> 34))
    assert n_multi > 200 and n_both > 20
    # the per-strand pass of the same handle (no direct table: plain table + rank steps + finish on the text) still answers
    want_keys, want_res = _expected(orc, hidx, reads, M, k + 5, 15, 1)
    keys, res = _run(amd, fmi, orc.pack4(flat), 4, R, M, k + 5, 15, 1)
    assert np.array_equal(keys, want_keys) and res == want_res
    # requests outside what the table serves are refused
    qs = amd.PackedStringSet(orc.pack4(flat), 4, R * 2, fixed_len=k + 8, stride=M, seeds_per_string=2, seed_interval=20)
    with pytest.raises(amd.NvbioError):
        fmi.match_seed_diagonals_both(qs, M)
    fmi.close()
    plain = amd.FMIndex.build(orc.pack2(text), n, kmer_len=k, sa_int=1)
    assert not plain.canonical
    qs = amd.PackedStringSet(orc.pack4(flat), 4, R * 2, fixed_len=k + 2, stride=M, seeds_per_string=2, seed_interval=20)
    with pytest.raises(amd.NvbioError):
        plain.match_seed_diagonals_both(qs, M)
    plain.close()
    with pytest.raises(amd.NvbioError):
        amd.FMIndex.build(orc.pack2(text), n, kmer_len=8, sa_int=1, table_flags=amd.FM_TABLE_CANONICAL)      # even k


def _all_candidates(hidx, keys, res_f, res_r, spr, M, L, S, cap=None):
    """every candidate key a run of the two-strand pass stands for: its keys plus the rows of its residual ranges (first `cap` rows)"""
    out = [np.asarray(keys, dtype=np.int64)]
    for strand, res in ((0, res_f), (1, res_r)):
        for sid, (x, y) in res.items():
            if cap is not None:
                y = min(y, x + cap - 1)
            rid, j = sid // spr, sid % spr
            p = (M - j * S - L) if strand else j * S
            out.append((np.int64(rid) << 34) | (strand << 33) | (hidx.sa[x:y + 1].astype(np.int64) + 1024 - p))
    return np.unique(np.concatenate(out))


@pytest.mark.parametrize("k", [5, 9, 15])
def test_two_strand_pass_with_deferred_heavy_searches(amd, orc, k):
    """NVBIO_FM_DEFER_HEAVY: the searches the canonical table cannot answer (k-mers with more than 8 occurrences -- all of them at k = 5 --
    or more hits on a strand than the in-line limit) run as a dense launch behind the pass.  The candidates the call stands for
    (keys + rows of the residual ranges) are those of the in-line form, and the keys that do not depend on a deferred search are where
    they were: the pass's own keys, tile by tile; the deferred searches' keys follow them."""
    rng = np.random.default_rng(1200 + k)
    n = 300007
    text = _text(rng, n)
    text[160000:160300] = 3 - text[5000:5300][::-1]
    hidx = orc.build_index(text)
    fmi = amd.FMIndex.build(orc.pack2(text), n, kmer_len=k, sa_int=1, table_flags=amd.FM_TABLE_CANONICAL_WIDE)
    R, M = 3000, 150
    reads = _reads(rng, text, R, M)
    packed = orc.pack4(reads.reshape(-1))
    n_deferred = 0
    for L, S in ((k + 5, 15), (k + 7, 20)):
        spr = (M - L) // S + 1
        want_f, res_f = _expected(orc, hidx, reads, M, L, S, 0)
        want_r, res_r = _expected(orc, hidx, reads, M, L, S, 1)
        want_all = _all_candidates(hidx, np.concatenate([want_f, want_r]), res_f, res_r, spr, M, L, S)
        for h in (0, 4):
            for gb in (0, 64):
                qs = amd.PackedStringSet(packed, 4, R * spr, fixed_len=L, stride=M, seeds_per_string=spr, seed_interval=S)
                b = fmi.match_seed_diagonals_both(qs, M, grid_blocks=gb, inline_hits=h, defer_heavy=True)
                c = [int(v) for v in b["counts"][:4].cpu().numpy()]
                keys = b["keys"][:c[0]].cpu().numpy()
                nq = R * spr
                res = []
                for lo, cnt in ((0, c[1]), (nq, c[2])):
                    rr = amd.u32(b["ranges"][lo:lo + cnt]); ids = b["ids"][lo:lo + cnt].cpu().numpy()
                    assert len(np.unique(ids)) == len(ids)
                    res.append({int(i): (int(a), int(d)) for i, (a, d) in zip(ids, rr)})
                got_all = _all_candidates(hidx, keys, res[0], res[1], spr, M, L, S)
                assert np.array_equal(got_all, want_all), (k, L, S, h, gb)
                # residual entries are real multi-row ranges of the oracle's searches
                for strand, want_res in ((0, res_f), (1, res_r)):
                    for sid, xy in res[strand].items():
                        assert want_res[sid] == xy
                n_deferred += c[1] + c[2]
    assert n_deferred > 500
    fmi.close()


def test_two_strand_pass_over_ragged_reads(amd, orc):
    """reads of different lengths, each seeded at its own interval (nvBowtie: seed_freq( read_len ), mapping_inl.h:507-529): the pass over
    a ragged seed set (seed_intervals_dev) leaves, read by read, what the uniform pass leaves for a batch of that read's length; seed ids
    beyond a read's last seed match nothing -- in the two-strand pass and in the plain match()"""
    rng = np.random.default_rng(4242)
    n = 300007
    text = _text(rng, n)
    hidx = orc.build_index(text)
    k, L = 11, 16
    fmi = amd.FMIndex.build(orc.pack2(text), n, kmer_len=k, sa_int=1, table_flags=amd.FM_TABLE_CANONICAL_WIDE)
    R, Mmax = 2500, 150
    full = _reads(rng, text, R, Mmax)
    lens = rng.integers(60, Mmax + 1, R); lens[:40] = rng.integers(10, 25, 40)              # some reads shorter than two seeds, some than one
    lens[40:60] = Mmax
    offs = np.zeros(R + 1, dtype=np.int64); offs[1:] = np.cumsum(lens)
    flat = np.concatenate([full[r, :lens[r]] for r in range(R)])
    itab = np.maximum((np.float32(1.0) + np.float32(1.15) * np.sqrt(np.arange(Mmax + 1, dtype=np.float32))).astype(np.int32), 1)
    ivs = itab[lens]
    nseeds = np.where(lens >= L, (lens - L) // ivs + 1, 0)
    spr = int(nseeds.max())
    qs = amd.PackedStringSet(orc.pack4(flat), 4, R * spr, offsets=offs.astype(np.uint32), fixed_len=L, stride=0, seeds_per_string=spr,
                             seed_intervals=ivs.astype(np.uint32))
    # expectations, length group by length group, from the uniform helper
    exp_keys, exp_res = [[], []], [{}, {}]
    for ln in np.unique(lens):
        grp = np.nonzero(lens == ln)[0]
        if ln < L:
            continue
        S = int(itab[ln]); g_spr = (ln - L) // S + 1
        sub = np.stack([full[r, :ln] for r in grp])
        for strand in (0, 1):
            ks, rs = _expected(orc, hidx, sub, int(ln), L, S, strand)
            lrid = ks >> 34
            exp_keys[strand].append((grp[lrid].astype(np.int64) << 34) | (ks & ((1 << 34) - 1)))
            for sid, xy in rs.items():
                exp_res[strand][int(grp[sid // g_spr]) * spr + sid % g_spr] = xy
    for defer in (False, True):
        b = fmi.match_seed_diagonals_both(qs, Mmax, defer_heavy=defer)
        c = [int(v) for v in b["counts"][:4].cpu().numpy()]
        keys = b["keys"][:c[0]].cpu().numpy()
        nq = R * spr
        res = []
        for lo, cnt in ((0, c[1]), (nq, c[2])):
            rr = amd.u32(b["ranges"][lo:lo + cnt]); ids = b["ids"][lo:lo + cnt].cpu().numpy()
            res.append({int(i): (int(a), int(d)) for i, (a, d) in zip(ids, rr)})
        if not defer:
            for strand in (0, 1):
                got = np.sort(keys[((keys >> 33) & 1) == strand])
                assert np.array_equal(got, np.sort(np.concatenate(exp_keys[strand]))), strand
                assert res[strand] == exp_res[strand], strand
        else:
            # (the deferred searches' keys come without the adjacent-duplicate removal: compare what the call stands for)
            def cands(keys, res):
                out = [keys]
                for strand in (0, 1):
                    for sid, (x, y) in res[strand].items():
                        rid, j = sid // spr, sid % spr
                        p = (int(lens[rid]) - j * int(ivs[rid]) - L) if strand else j * int(ivs[rid])
                        out.append((np.int64(rid) << 34) | (strand << 33) | (hidx.sa[x:y + 1].astype(np.int64) + 1024 - p))
                return np.unique(np.concatenate(out))
            assert np.array_equal(cands(keys, res), cands(np.concatenate(exp_keys[0] + exp_keys[1]), exp_res))
    # the plain operator over the same ragged seed set: ranges of the seeds that exist, (1, 0) for the ids that do not
    ranges = amd.u32(fmi.match(qs))
    sid = np.arange(R * spr); rid, j = sid // spr, sid % spr
    exists = j < nseeds[rid]
    assert (ranges[~exists, 0] > ranges[~exists, 1]).all()
    beg = offs[rid] + j * ivs[rid]
    sel = np.nonzero(exists)[0][:4000]
    syms = np.concatenate([flat[b:b + L] for b in beg[sel]])
    want = orc.match_batch(hidx, syms, (np.arange(len(sel) + 1) * L).astype(np.uint32))
    assert np.array_equal(ranges[sel], want)
    assert exists.sum() > 10000 and (~exists).sum() > 500
    fmi.close()
