"""GPU, BASELINE.json's full sizes, through size-independent properties (the oracle cannot build or
hold a 3 Gbp index in seconds): config 2 (1 M x 22 bp seeds vs a 3 Gbp synthetic reference) and a
6.25 M-pair slice of config 4 (band-31 local Gotoh, 150 bp) -- one GPU's share of the 50 M pairs."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ref3g(amd):
    import torch
    n = 3_000_000_000
    g = torch.Generator(device="cuda:0"); g.manual_seed(99)
    words = torch.randint(-2 ** 31, 2 ** 31 - 1, ((n + 15) // 16 + 8,), dtype=torch.int64, device="cuda:0", generator=g).to(torch.int32)
    fmi = amd.FMIndex.build(words, n, kmer_len=12, sa_int=16)
    yield n, words, fmi
    fmi.close()


def _symbols(words, idx):
    import torch
    w = words[idx >> 4].to(torch.int64) & 0xFFFFFFFF
    return ((w >> (30 - 2 * (idx & 15))) & 3).to(torch.uint8)


def test_config2_seeds_vs_3gbp(amd, ref3g):
    import torch
    n, words, fmi = ref3g
    v = fmi.view()
    assert v.length == n and v.L2[4] == n and 0 < v.primary <= n
    counts = [v.L2[c + 1] - v.L2[c] for c in range(4)]
    assert all(abs(c - n / 4) < 1e6 for c in counts)                # i.i.d. text
    Q, L = 1_000_000, 22
    g = torch.Generator(device="cuda:0"); g.manual_seed(7)
    starts = torch.randint(0, n - L, (Q,), device="cuda:0", generator=g)
    sym = _symbols(words, starts[:, None] + torch.arange(L, device="cuda:0")[None, :]).contiguous()
    sym[900_000:] = torch.randint(0, 4, (100_000, L), device="cuda:0", generator=g, dtype=torch.uint8)   # 10 % random: mostly misses
    qs = amd.PackedStringSet(sym.view(-1), 8, Q, fixed_len=L)
    r = fmi.match(qs)
    r0, blocks = fmi.match(qs, amd.FM_NO_KMER_TABLE, want_blocks=True)
    assert torch.equal(r, r0)                                       # the k-mer table changes nothing, hits or misses
    ru = r.to(torch.int64) & 0xFFFFFFFF
    hit = ru[:, 0] <= ru[:, 1]
    assert bool(hit[:900_000].all())                                # every substring of the text is found
    assert int(hit[900_000:].sum()) < 1000                          # random 22-mers: 3e9 / 4^22 expected hit rate
    b = blocks.to(torch.int64)
    assert int(b.min()) >= 1 and int(b.max()) <= 2 * L              # 704..1408 B per query
    assert 25 < float(b[:900_000].float().mean()) < 40              # ~13 steps x 2 blocks + 9 x 1
    # locate every hit row: the positions spell the query, and almost always are the sampled position
    rows = ru[:900_000, 0].to(torch.int32).contiguous()
    pos = fmi.locate(rows).to(torch.int64) & 0xFFFFFFFF
    assert bool((pos < n).all())
    assert torch.equal(_symbols(words, pos[:, None] + torch.arange(L, device="cuda:0")[None, :]), sym[:900_000])
    assert float((pos == starts[:900_000]).float().mean()) > 0.999
    # LF walk consistency: inv_psi moves one symbol to the left
    prev = fmi.basic_inv_psi(rows)
    pos_prev = fmi.locate(prev).to(torch.int64) & 0xFFFFFFFF
    ok = pos > 0
    assert torch.equal(pos_prev[ok], pos[ok] - 1)
    # filter: scan of range sizes and expansion are consistent
    flt = amd.FMIndexFilter()
    total = flt.rank(fmi, qs)
    assert total == int((ru[:, 1] + 1 - ru[:, 0]).clamp(min=0).sum())
    hits = flt.locate(0, total)
    assert int(hits[:, 1].max()) < Q and bool((hits[1:, 1] >= hits[:-1, 1]).all())      # grouped by query, in order


def test_config4_band31_local_slice(amd, ref3g):
    import torch
    n, words, fmi = ref3g
    P, M = 6_250_000, 150
    g = torch.Generator(device="cuda:0"); g.manual_seed(8)
    starts = torch.randint(16, n - M - 64, (P,), device="cuda:0", generator=g)
    sym = _symbols(words, starts[:, None] + torch.arange(M, device="cuda:0")[None, :])
    # plant exactly e substitutions per read (e = pair index mod 4) far from each other and from the ends
    e = torch.arange(P, device="cuda:0") % 4
    for k in range(3):
        col = 30 + 40 * k
        m = e > k
        sym[m, col] = (sym[m, col] + 1) % 4
    s = sym.reshape(-1, 8).to(torch.int64)
    sh = torch.tensor([28, 24, 20, 16, 12, 8, 4, 0], device="cuda:0")
    w = (s << sh[None, :]).sum(dim=1)
    reads4 = torch.cat([torch.where(w >= 2 ** 31, w - 2 ** 32, w).to(torch.int32), torch.zeros(4, dtype=torch.int32, device="cuda:0")])
    roffs = (torch.arange(P + 1, device="cuda:0") * M).to(torch.int32)
    wb = (starts - 15).to(torch.int32)
    we = (starts - 15 + M + 31).to(torch.int32)
    batch = amd.AlignmentBatch(reads4, 4, roffs, words, 2, wb, we, max_read_len=M)
    al = amd.make_gotoh_aligner(amd.LOCAL, amd.GotohScheme(2, 6, 6, -8, -3, -8, -3))
    sc, sk = amd.batch_banded_alignment_score(31, al, batch)
    # an isolated substitution costs 2 + 6 against an otherwise perfect 150 bp diagonal: score = 300 - 8 e,
    # ending at the last cell of the diagonal: sink = (15 + 150, 150)
    assert torch.equal(sc, (300 - 8 * e).to(torch.int32))
    sku = sk.to(torch.int64) & 0xFFFFFFFF
    assert bool((sku[:, 0] == 165).all()) and bool((sku[:, 1] == 150).all())
    # the int32 kernel (no max_read_len hint) agrees on a slice
    sub = amd.AlignmentBatch(reads4, 4, roffs, words, 2, wb[:200_000], we[:200_000])
    sc2, sk2 = amd.batch_banded_alignment_score(31, al, sub)
    assert torch.equal(sc2, sc[:200_000]) and torch.equal(sk2, sk[:200_000])


# ---------------------------------------------------------------------------------------------------------------------
# The headline configuration of bench.py (BASELINE.json configs[2]): 3 Gbp reference, k = 17 direct table (34-bit keys, 128 GiB,
# groups), full suffix array (sa_int = 1), one-call seed pass -- checked against the reference's algorithm run by the plain
# operators on the same index (match() without any table + locate()), which the small-scale tests pin on the reference's
# own outputs.  The shape of nvbio-test/fmindex_test.cu:603-709: match -> locate -> compare, here over 1 M reads x 18 seeds.
# ---------------------------------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def headline(amd):
    import importlib
    import torch
    import bench
    n = 3_000_000_000
    genome = bench.make_reference(n, "cuda:0", seed=1234)
    try:
        fmi = amd.FMIndex.build(genome, n, kmer_len=17, sa_int=1)
    except amd.NvbioError as e:
        pytest.skip("the k = 17 tables do not fit on this device: %s" % e)
    R, M = 1_000_000, 150
    reads_sym, truth_pos, truth_rc = bench.make_reads(genome, n, R, M, "cuda:0", seed=77)
    reads_sym[5::1000, 40] = 4                                  # a few N's
    reads4 = bench.pack4(reads_sym.view(-1))
    pipeline = importlib.import_module("nvbio_gpl_amd.pipeline")
    yield dict(n=n, genome=genome, fmi=fmi, R=R, M=M, reads4=reads4, truth_pos=truth_pos, truth_rc=truth_rc, pipeline=pipeline)
    fmi.close()
    torch.cuda.empty_cache()


def test_headline_seed_pass_equals_reference_algorithm(amd, headline):
    """fm_seed_tiles_kernel over the k = 17 direct table == match() through the reference's algorithm (no table) + locate(),
    key for key and in order, and seed for seed on the residual list, both strands"""
    import torch
    h = headline
    fmi, R, M = h["fmi"], h["R"], h["M"]
    L, S = 22, 15
    spr = (M - L) // S + 1
    qs = amd.PackedStringSet(h["reads4"], 4, R * spr, fixed_len=L, stride=M, seeds_per_string=spr, seed_interval=S)
    v = fmi.view()
    assert v.sa_int == 1 and fmi.supports_direct() and fmi.device_bytes() > 150e9      # 32 + 128 GiB of tables, full SA
    sid = torch.arange(R * spr, device="cuda:0", dtype=torch.int64)
    n_single = n_multi = 0
    for strand, flags in ((0, 0), (1, amd.FM_SCAN_FORWARD | amd.FM_COMPLEMENT)):
        ref = fmi.match(qs, flags | amd.FM_NO_KMER_TABLE)                              # the reference's algorithm, symbol by symbol
        assert torch.equal(fmi.match(qs, flags), ref)                                  # ... the k = 16 plain table changes nothing
        ru = ref.to(torch.int64) & 0xFFFFFFFF
        x, y = ru[:, 0], ru[:, 1]
        single, multi = x == y, x < y
        pos = fmi.locate(x[single].to(torch.int32)).to(torch.int64) & 0xFFFFFFFF
        rid, j = sid[single] // spr, sid[single] % spr
        p = j * S
        if strand:
            p = M - p - L
        keys = (rid << 34) | (strand << 33) | (pos + 1024 - p)
        keep = torch.ones_like(keys, dtype=torch.bool)
        keep[1:] = keys[1:] != keys[:-1]
        want = keys[keep]
        b = fmi.match_seed_diagonals(qs, flags, M, strand)
        nk, nr = [int(c) for c in b["counts"][:2].cpu()]
        assert nk == want.numel() and torch.equal(b["keys"][:nk], want)
        # residual seeds: exactly the multi-row ones, with the reference's ranges
        order = torch.argsort(b["ids"][:nr])
        assert torch.equal(b["ids"][:nr][order].to(torch.int64), sid[multi])
        assert torch.equal(b["ranges"][:nr][order], ref[multi])
        # match_direct agrees too: sizes, and positions where it finished on the text
        rng_d, direct = fmi.match_direct(qs, flags)
        du = rng_d.to(torch.int64) & 0xFFFFFFFF
        d = direct.bool()
        assert torch.equal((du[:, 1] + 1 - du[:, 0]).clamp(min=0), (y + 1 - x).clamp(min=0))
        assert bool((single[d]).all()) and torch.equal(du[d, 0], fmi.locate(x[d].to(torch.int32)).to(torch.int64) & 0xFFFFFFFF)
        n_single += int(single.sum()); n_multi += int(multi.sum())
        del ref, ru, b
    assert n_single > 3 * R and 0 < n_multi < n_single // 100      # ~40 % of 18 M seeds hit once; a handful of true repeats


def test_headline_pipeline_three_ways(amd, headline):
    """seed_and_extend on the headline index: one-call seed pass == separate direct operators == plain match() + locate(),
    and every planted read comes back to its locus and strand"""
    import torch
    h = headline
    pipeline, fmi = h["pipeline"], h["fmi"]
    batch = pipeline.ReadBatch(h["reads4"], h["R"], h["M"])
    outs = []
    for fused, direct in ((True, True), (False, True), (False, False)):
        params = pipeline.SeedExtendParams.end_to_end()
        params.fused_seed_pass, params.direct = fused, direct
        outs.append(pipeline.seed_and_extend(fmi, h["genome"], h["n"], batch, params))
    (bs, bp, brc, nc) = outs[0]
    for o in outs[1:]:
        assert torch.equal(o[0], bs) and torch.equal(o[1], bp) and torch.equal(o[2], brc)
        assert abs(o[3] - nc) <= nc // 100                        # the two routes drop slightly different duplicates (a repeat's hits
                                                                  # are deduplicated in line by one, on their own list by the other)
    params = pipeline.SeedExtendParams.end_to_end()
    aligned = bs >= params.min_score_for(h["M"])
    near = (bp - (h["truth_pos"] + h["M"])).abs() <= 40
    assert float(aligned.float().mean()) > 0.999
    assert float((aligned & near & (brc.bool() == h["truth_rc"])).float().mean()) > 0.999
    # the DP-for-everything variant of the extension agrees (nvbio_alignment_batch::algo_flags)
    params.algo_flags = amd.ALN_NO_UNGAPPED_SCORE
    o = pipeline.seed_and_extend(fmi, h["genome"], h["n"], batch, params)
    assert torch.equal(o[0], bs) and torch.equal(o[1], bp) and torch.equal(o[2], brc)
