"""GPU, full size: the configuration bench.py times since round 2 -- 3 Gbp reference, the CANONICAL k = 17 table (2^33 entries of 16 bytes,
128 GiB: a k-mer and its reverse complement share an entry that holds up to two occurrences), full suffix array, ONE seed pass for both strands -- checked against the
reference's algorithm run by the plain operators on the same index (match() without any table + locate(), which the small-scale tests
pin on the reference's own outputs).  The shape of nvbio-test/fmindex_test.cu:603-709: match -> locate -> compare, over 1 M reads x 9
seed windows x 2 strands.  (Its own module: the direct-table handle of test_gpu_fullsize.py has to be released first.)"""
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def headline(amd):
    import importlib
    import torch
    import bench
    torch.cuda.empty_cache()
    n = 3_000_000_000
    genome = bench.make_reference(n, "cuda:0", seed=1234)
    try:
        fmi = amd.FMIndex.build(genome, n, kmer_len=17, sa_int=1, table_flags=amd.FM_TABLE_CANONICAL_WIDE)
    except amd.NvbioError as e:
        pytest.skip("the k = 17 canonical table does not fit on this device: %s" % e)
    R, M = 1_000_000, 150
    reads_sym, truth_pos, truth_rc = bench.make_reads(genome, n, R, M, "cuda:0", seed=78)
    reads_sym[5::1000, 40] = 4                                  # a few N's
    reads_sym[7::5000] = reads_sym[8::5000]                     # a few duplicated reads
    reads4 = bench.pack4(reads_sym.view(-1))
    pipeline = importlib.import_module("nvbio_gpl_amd.pipeline")
    yield dict(n=n, genome=genome, fmi=fmi, R=R, M=M, reads4=reads4, truth_pos=truth_pos, truth_rc=truth_rc, pipeline=pipeline)
    fmi.close()
    torch.cuda.empty_cache()


def test_two_strand_seed_pass_equals_reference_algorithm(amd, headline):
    """fm_seed_both_kernel over the canonical table == match() of every seed and of its reverse complement through the reference's
    algorithm (no table) + locate(): the keys tile by tile (forward strand, then reverse strand, each in seed order), the residual lists
    seed for seed; with NVBIO_FM_INLINE_HITS the same candidates as a multiset"""
    import torch
    h = headline
    fmi, R, M = h["fmi"], h["R"], h["M"]
    L, S = 22, 15
    spr = (M - L) // S + 1
    rpt = 64 // spr
    qs = amd.PackedStringSet(h["reads4"], 4, R * spr, fixed_len=L, stride=M, seeds_per_string=spr, seed_interval=S)
    assert fmi.canonical and fmi.view().sa_int == 1 and 170e9 < fmi.device_bytes() < 200e9     # 32 + 128 GiB of tables, full SA
    sid = torch.arange(R * spr, device="cuda:0", dtype=torch.int64)
    b = fmi.match_seed_diagonals_both(qs, M)
    c = [int(v) for v in b["counts"][:4].cpu()]
    got = b["keys"][:c[0]].clone()
    want, multis, expanded = [], [], []
    for strand, flags in ((0, 0), (1, amd.FM_SCAN_FORWARD | amd.FM_COMPLEMENT)):
        ref = fmi.match(qs, flags | amd.FM_NO_KMER_TABLE)                              # the reference's algorithm, symbol by symbol
        assert torch.equal(fmi.match(qs, flags), ref)                                  # ... the k = 16 plain table changes nothing
        ru = ref.to(torch.int64) & 0xFFFFFFFF
        x, y = ru[:, 0], ru[:, 1]
        single, multi = x == y, x < y
        pos = fmi.locate(x[single].to(torch.int32)).to(torch.int64) & 0xFFFFFFFF
        rid, j = sid[single] // spr, sid[single] % spr
        p = (M - j * S - L) if strand else j * S
        keys = (rid << 34) | (strand << 33) | (pos + 1024 - p)
        keep = torch.ones_like(keys, dtype=torch.bool)
        keep[1:] = keys[1:] != keys[:-1]
        want.append(keys[keep])
        # residual seeds of this strand: exactly the multi-row ones, with the reference's ranges
        lo, cnt = (R * spr, c[2]) if strand else (0, c[1])
        order = torch.argsort(b["ids"][lo:lo + cnt])
        assert torch.equal(b["ids"][lo:lo + cnt][order].to(torch.int64), sid[multi])
        assert torch.equal(b["ranges"][lo:lo + cnt][order], ref[multi])
        multis.append(int(multi.sum()))
        # the keys those seeds would add (all rows of ranges of up to 4 rows), for the inline variant below
        rows = (y - x + 1)[multi]
        small = rows <= 4
        xs, ids_s, rows_s = x[multi][small], sid[multi][small], rows[small]
        rep = torch.repeat_interleave(torch.arange(xs.numel(), device="cuda:0"), rows_s)
        first = torch.cumsum(rows_s, 0) - rows_s
        r_all = xs[rep] + (torch.arange(rep.numel(), device="cuda:0") - first[rep])
        pos_m = fmi.locate(r_all.to(torch.int32)).to(torch.int64) & 0xFFFFFFFF
        rid_m, j_m = ids_s[rep] // spr, ids_s[rep] % spr
        p_m = (M - j_m * S - L) if strand else j_m * S
        expanded.append(((rid_m << 34) | (strand << 33) | (pos_m + 1024 - p_m), int((~small).sum())))
        del ref, ru
    # tile-major order: sort the expectation by (tile, strand), stable within
    allk = torch.cat(want)
    order_key = ((allk >> 34) // rpt) * 2 + ((allk >> 33) & 1)
    allk = allk[torch.sort(order_key, stable=True).indices]
    assert c[0] == allk.numel() and torch.equal(got, allk)
    assert allk.numel() > R and 0 < sum(multis) < R // 10          # one candidate per read after the duplicate removal; a handful of true repeats
    # NVBIO_FM_INLINE_HITS(4): the short repeats leave their keys directly; what is left on the residual lists are the larger ranges
    b = fmi.match_seed_diagonals_both(qs, M, inline_hits=4)
    c = [int(v) for v in b["counts"][:4].cpu()]
    assert (c[1], c[2]) == (expanded[0][1], expanded[1][1])
    want_all = torch.sort(torch.cat([allk, expanded[0][0], expanded[1][0]])).values
    assert torch.equal(torch.sort(b["keys"][:c[0]]).values, want_all)


def test_pipeline_over_the_canonical_table(amd, headline):
    """seed_and_extend: the two-strand pass == one pass per strand on the same handle (plain table + rank steps) == plain match() +
    locate(); every planted read comes back to its locus and strand"""
    import torch
    h = headline
    pipeline, fmi = h["pipeline"], h["fmi"]
    batch = pipeline.ReadBatch(h["reads4"], h["R"], h["M"])
    outs = []
    for two, fused, direct in ((True, True, True), (False, True, True), (False, False, False)):
        params = pipeline.SeedExtendParams.end_to_end()
        params.two_strand_pass, params.fused_seed_pass, params.direct = two, fused, direct
        timers = {}
        outs.append(pipeline.seed_and_extend(fmi, h["genome"], h["n"], batch, params, timers))
        assert ("match_both" in timers) == two
    (bs, bp, brc, nc) = outs[0]
    for o in outs[1:]:
        assert torch.equal(o[0], bs) and torch.equal(o[1], bp) and torch.equal(o[2], brc)
        assert abs(o[3] - nc) <= nc // 100
    params = pipeline.SeedExtendParams.end_to_end()
    aligned = bs >= params.min_score_for(h["M"])
    near = (bp - (h["truth_pos"] + h["M"])).abs() <= 40
    ok = aligned & near & (brc.bool() == h["truth_rc"])
    ok[7::5000] = True                                          # the overwritten reads carry another read's truth
    assert float(aligned.float().mean()) > 0.999 and float(ok.float().mean()) > 0.999
