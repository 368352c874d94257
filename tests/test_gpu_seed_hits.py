"""GPU parity of nvBowtie's seed-hit bookkeeping kernels (nvbio_seed_hits_map / _select / nvbio_score_reduce_effort) against the
oracle: the deque arrays hit for hit (the interval heap's own order: pinned on the reference's priority_deque through the
oracle, tests/golden/deque_golden.npz), the rows select hands out, and the whole best-approx loop with nvBowtie's choices --
max_hits cap, smallest range first, effort counter, reseeding -- read for read."""
import importlib

import numpy as np
import pytest

import oracle
from oracle import cpu_pipeline
from util import mutate_reads

pytestmark = pytest.mark.gpu


def test_map_and_select_equal_the_oracle(amd, orc):
    import torch
    rng = np.random.default_rng(12)
    R, spr, M, L, S = 3000, 9, 150, 22, 15
    for max_hits in (100, 5, 1):
        # random match results: mostly one-row ranges, some empty, some wide (ties everywhere, as in real data)
        def ranges():
            x = rng.integers(0, 3_000_000_000, (R, spr))
            size = rng.choice([0, 0, 1, 1, 1, 1, 2, 3, 40, 5000], (R, spr))
            return np.stack([np.where(size == 0, 1, x), np.where(size == 0, 0, x + size - 1)], axis=2).astype(np.uint32)
        fw, rc = ranges(), ranges()
        sp = amd.SeedHitsParams(spr, S, L, M, max_hits=max_hits, rep_seeds=300)
        cap = sp.capacity()
        deques = torch.zeros((R, cap, 2), dtype=torch.int32, device="cuda:0")
        sizes = torch.zeros(R, dtype=torch.int32, device="cuda:0"); reseed = torch.zeros(R, dtype=torch.uint8, device="cuda:0")
        amd.seed_hits_map(torch.from_numpy(fw.view(np.int32)).cuda(), torch.from_numpy(rc.view(np.int32)).cuda(), sp, R, deques, sizes, reseed)
        d, n, rs = amd.u32(deques), sizes.cpu().numpy(), reseed.cpu().numpy()
        want = [orc.map_exact_read(fw[r], rc[r], np.arange(spr) * S, M, L, max_hits, 300) for r in range(R)]
        for r in range(R):
            assert n[r] == len(want[r][0]) and np.array_equal(d[r, :n[r]], want[r][0]) and bool(rs[r]) == want[r][1], (max_hits, r)
        # select until every deque is empty (wide ranges capped at 30 passes): rows, packed seeds and flags per read equal the oracle's
        state = [(w[0].copy(), 1) for w in want]
        active = (torch.arange(R, dtype=torch.int32, device="cuda:0") | (1 << 31)).contiguous()      # top flag set, as with --top
        count = torch.zeros(1, dtype=torch.int32, device="cuda:0")
        for p in range(30):
            na = active.numel()
            if na == 0:
                break
            hits = amd.HitQueues(torch.empty(na, dtype=torch.int32, device="cuda:0"), torch.empty(na, dtype=torch.int32, device="cuda:0"),
                                 torch.empty(na, dtype=torch.int32, device="cuda:0"))
            out = torch.empty(na, dtype=torch.int32, device="cuda:0")
            amd.seed_hits_select(active, None, sp, deques, sizes, hits, out, count)
            nh = int(count.item())
            got = {}
            a_out = amd.u32(out[:nh]); hr = amd.u32(hits.read_id[:nh]); hl = amd.u32(hits.loc[:nh]); hs = amd.u32(hits.seed[:nh])
            for k in range(nh):
                assert (a_out[k] & 0x7FFFFFFF) == hr[k] and (a_out[k] >> 31) == ((hs[k] >> 14) & 1)
                got[int(hr[k])] = (int(hl[k]), int(hs[k]))
            exp = {}
            for r in (amd.u32(active) & 0x7FFFFFFF):
                dq, top = state[r]
                ok, row, seed, top, dq = orc.select_read(dq, top)
                state[r] = (dq, top)
                if ok:
                    exp[int(r)] = (row, seed)
            assert got == exp, (max_hits, p)
            active = out[:nh].contiguous()


@pytest.mark.parametrize("mode", ["default", "tight"])
def test_best_approx_loop_equals_the_oracle(amd, orc, mode):
    """the whole loop: deques, select, locate, BestScoreStream window + orientation over reads stored reversed, banded DP, the
    arrival-order best / second best with the effort counter, reseeding -- best and second alignment of every read and the number
    of extensions equal the oracle's read-by-read restatement"""
    import torch
    pipeline = importlib.import_module("nvbio_gpl_amd.pipeline")
    rng = np.random.default_rng(31)
    G = 400_000
    text = rng.integers(0, 4, G, dtype=np.uint8)
    unit = rng.integers(0, 4, 250, dtype=np.uint8)
    for c in range(30):                                            # a 30-copy repeat: wide ranges, the cap, the effort limit
        text[50000 + 4000 * c:50250 + 4000 * c] = unit
        text[50000 + 4000 * c + int(rng.integers(0, 250))] = rng.integers(0, 4)
    hidx = orc.build_index(text)
    R, M = 700, 150
    starts = rng.integers(0, G - M - 8, R)
    starts[:200] = 50000 + 4000 * rng.integers(0, 30, 200) + rng.integers(0, 90, 200)
    starts[200:210] = rng.integers(0, 6, 10); starts[210:220] = G - M - 8 - rng.integers(0, 4, 10)
    reads = mutate_reads(rng, text, starts, M, sub=0.03)
    rcm = rng.random(R) < 0.5
    reads[rcm] = 3 - reads[rcm][:, ::-1]
    reads[rng.random(reads.shape) < 0.002] = 4
    reads[-15:] = rng.integers(0, 4, (15, M))                      # unalignable: no hits, reseeding
    kw = dict(max_hits=100, rep_seeds=1000, max_effort=15, min_ext=30, max_ext=400, max_reseed=2) if mode == "default" else \
         dict(max_hits=6, rep_seeds=8, max_effort=2, min_ext=3, max_ext=12, max_reseed=2)
    params = pipeline.SeedExtendParams.end_to_end()
    osc = oracle.Scheme(0, 6, 6, -8, -3, -8, -3)
    want = cpu_pipeline.nvbowtie_best_approx_cpu(orc, hidx, text, G, reads, osc, oracle.SEMI_GLOBAL, params.min_score_for(M), **kw)

    genome2 = orc.pack2(text)
    fmi = amd.FMIndex.build(genome2, G, kmer_len=8, sa_int=16)
    stored = np.ascontiguousarray(reads[:, ::-1])
    rb = pipeline.ReadBatch(torch.from_numpy(orc.pack4(stored.reshape(-1)).view(np.int32)).cuda(), R, M)
    g_dev = torch.from_numpy(genome2.view(np.int32)).cuda()
    got = pipeline.nvbowtie_best_approx(fmi, g_dev, G, rb, params, pipeline.NvBowtieParams(**kw))
    for k in ("best_score", "best_loc", "best_rc", "second_score", "second_loc", "second_rc"):
        assert np.array_equal(got[k].cpu().numpy().astype(np.int64), want[k].astype(np.int64)), (mode, k)
    assert got["n_extensions"] == want["n_extensions"]
    aligned = got["best_loc"].cpu().numpy() >= 0
    assert aligned[:-15].mean() > 0.97 and not aligned[-15:].any()
    # the same loop as the C++ host loop over the C ABI (lib/libnvbio_amd_host.so), one hit per read and pass: identical
    host = pipeline.nvbowtie_best_approx_host(fmi, g_dev, G, rb, params, pipeline.NvBowtieParams(**kw), multi_hit=False)
    for k in ("best_score", "best_loc", "best_rc", "second_score", "second_loc", "second_rc"):
        assert np.array_equal(host[k].cpu().numpy().astype(np.int64), want[k].astype(np.int64)), (mode, k, "host")
    assert host["n_extensions"] == want["n_extensions"] and host["multi_passes"] == 0
    # ... and with the reference's several-hits-per-read phase (active reads <= BATCH_SIZE / 2: n = BATCH_SIZE / active hits per read and pass),
    # against the oracle's pass-by-pass restatement; two batch sizes, so that the phase starts at different points of the loop
    for bs in (0, 3 * R):
        wantb = cpu_pipeline.nvbowtie_best_approx_batch_cpu(orc, hidx, text, G, reads, osc, oracle.SEMI_GLOBAL, params.min_score_for(M), batch_size=bs or None, **kw)
        hostb = pipeline.nvbowtie_best_approx_host(fmi, g_dev, G, rb, params, pipeline.NvBowtieParams(**kw), batch_size=bs, multi_hit=True)
        for k in ("best_score", "best_loc", "best_rc", "second_score", "second_loc", "second_rc"):
            assert np.array_equal(hostb[k].cpu().numpy().astype(np.int64), wantb[k].astype(np.int64)), (mode, k, "multi", bs)
        assert hostb["n_extensions"] == wantb["n_extensions"] and hostb["passes"] == wantb["passes"] and hostb["multi_passes"] == wantb["multi_passes"]
        assert hostb["multi_passes"] > 0
    fmi.close()


@pytest.mark.parametrize("mode", ["default", "tight", "no_unpaired_ff"])
def test_paired_best_approx_loop_equals_the_oracle(amd, orc, mode):
    """the PAIRED-END loop (nvbio_host_best_approx_paired): anchor = mate 1 then mate 2, the anchor band-aligned against the pair-derived
    threshold (compute_target_score tightening as pairs are found), the opposite mate by full-matrix DP in its fragment window, the paired
    reduction with the unpaired fallback and the effort counter -- best_a / best_o of every pair, bit for bit, and the counters equal the
    oracle's pass-by-pass restatement; concordant pairs, pairs whose second mate lies elsewhere, a repeat family, unalignable mates"""
    import torch
    pipeline = importlib.import_module("nvbio_gpl_amd.pipeline")
    rng = np.random.default_rng(47)
    G = 300_000
    text = rng.integers(0, 4, G, dtype=np.uint8)
    unit = rng.integers(0, 4, 600, dtype=np.uint8)
    for c in range(12):                                            # a 12-copy repeat long enough to hold whole fragments
        text[40000 + 5000 * c:40600 + 5000 * c] = unit
        for _ in range(3):
            text[40000 + 5000 * c + int(rng.integers(0, 600))] = rng.integers(0, 4)
    hidx = orc.build_index(text)
    R, M1, M2 = 260, 100, 120
    frag = rng.integers(230, 480, R)
    starts = rng.integers(0, G - 500, R)
    starts[:60] = 40000 + 5000 * rng.integers(0, 12, 60) + rng.integers(0, 100, 60)
    starts[60:66] = rng.integers(0, 5, 6); starts[66:72] = G - frag[66:72] - rng.integers(0, 4, 6)
    ff = mode == "no_unpaired_ff"
    # FR: mate 1 forward at the fragment's start and mate 2 reverse-complemented at its end -- or, the fragment read from the other strand,
    # mate 1 reverse-complemented at the end and mate 2 forward at the start; FF: both forward, mate 1 upstream
    flip = (rng.random(R) < 0.5) & (not ff)
    p1 = np.where(flip, starts + frag - M1, starts)
    p2 = np.where(flip, starts, starts + frag - M2)
    p2[72:100] = rng.integers(0, G - M2, 28)                       # the second mate somewhere else: no concordant pair
    m1 = mutate_reads(rng, text, p1, M1, sub=0.03)
    m2 = mutate_reads(rng, text, p2, M2, sub=0.03)
    if not ff:
        m1[flip] = (3 - m1[flip][:, ::-1]).astype(np.uint8)
        m2[~flip] = (3 - m2[~flip][:, ::-1]).astype(np.uint8)
    m1[rng.random(m1.shape) < 0.002] = 4
    m1[-8:] = rng.integers(0, 4, (8, M1)); m2[-12:] = rng.integers(0, 4, (12, M2))          # unalignable mates
    kw = dict(max_hits=6, rep_seeds=8, max_effort=2, min_ext=3, max_ext=12, max_reseed=2) if mode == "tight" else \
         dict(max_hits=100, rep_seeds=1000, max_effort=15, min_ext=30, max_ext=400, max_reseed=2)
    policy = amd.PE_POLICY_FF if ff else amd.PE_POLICY_FR
    pe = pipeline.PairedEndParams(policy=policy, min_frag_len=0 if mode == "default" else 150, max_frag_len=500, overlap=mode != "tight")
    params = pipeline.SeedExtendParams.end_to_end()
    osc = oracle.Scheme(0, 6, 6, -8, -3, -8, -3)
    genome2 = orc.pack2(text)
    fmi = amd.FMIndex.build(genome2, G, kmer_len=8, sa_int=16)
    g_dev = torch.from_numpy(genome2.view(np.int32)).cuda()
    rb = [pipeline.ReadBatch(torch.from_numpy(orc.pack4(np.ascontiguousarray(m[:, ::-1]).reshape(-1)).view(np.int32)).cuda(), R, m.shape[1]) for m in (m1, m2)]
    n_paired = 0
    for bs, multi in ((0, False), (0, True), (3 * R, True)):
        want = cpu_pipeline.nvbowtie_best_approx_paired_cpu(orc, hidx, text, G, m1, m2, osc, oracle.SEMI_GLOBAL, params.min_score_for, batch_size=bs or None,
                                                            multi_hit=multi, policy=int(policy), min_frag=pe.min_frag_len, max_frag=pe.max_frag_len,
                                                            overlap=pe.overlap, unpaired=not ff, **kw)
        got = pipeline.nvbowtie_best_approx_paired_host(fmi, g_dev, G, rb[0], rb[1], params, pipeline.NvBowtieParams(**kw), pe, unpaired=not ff, batch_size=bs,
                                                        multi_hit=multi)
        for k in ("best_a", "best_o"):
            g = got[k].cpu().numpy().astype(np.int64)
            w = want[k]
            flags = w[..., 3] | (w[..., 4] << 1) | (w[..., 5] << 2)
            assert np.array_equal(g[..., 0], w[..., 0]), (mode, k, "score", bs, multi)
            assert np.array_equal(g[..., 1] & 0xFFFFFFFF, w[..., 1]), (mode, k, "pos", bs, multi)
            assert np.array_equal(g[..., 2] & 0xFFFFFFFF, w[..., 2]), (mode, k, "sink", bs, multi)
            assert np.array_equal(g[..., 3], flags), (mode, k, "flags", bs, multi)
        for k in ("n_extensions", "n_opposite", "passes"):
            assert got[k] == want[k], (mode, k, bs, multi)
        assert (got["multi_passes"] > 0) == multi
        a = got["best_a"].cpu().numpy()
        paired = ((a[:, 0, 3] >> 2) & 1) == 1
        n_paired = int(paired[:60].sum() + paired[100:-12].sum())
        assert paired[100:-12].mean() > (0.9 if mode == "default" else 0.75) and not paired[-12:].any()              # concordant pairs found as pairs; unalignable mates not
        if not ff:                                                                 # the per-mate fallback holds mate 1 of the pairs that are none
            assert (a[72:100, 0, 1] != -1).mean() > 0.9 and not paired[72:100].any()
    assert n_paired > 0
    # the traceback stage behind the loop: the anchor mate through the banded traceback in its band window, the opposite mate through the full-matrix
    # traceback from its window's begin to the column it ends in; CIGARs and alignment begins equal the oracle's tracebacks of the same strings
    tb = pipeline.nvbowtie_paired_traceback(g_dev, G, rb[0], rb[1], params, got, pipeline.NvBowtieParams(**kw), cigar_stride=32)
    ba = got["best_a"].cpu().numpy().astype(np.int64); bo = got["best_o"].cpu().numpy().astype(np.int64)
    is_pair = tb["paired"].cpu().numpy()
    assert np.array_equal(is_pair, (((ba[:, 0, 3] >> 2) & 1) == 1) & (ba[:, 0, 1] != -1))
    mates = (m1, m2)

    def oriented(read, rc):
        return (np.where(read[::-1] < 4, 3 - read[::-1], read[::-1]) if rc else read).astype(np.uint8)
    checked = 0
    for r in np.nonzero(is_pair)[0][::3]:
        am = int((ba[r, 0, 3] >> 1) & 1); a_rc = int(ba[r, 0, 3] & 1); o_rc = int(bo[r, 0, 3] & 1)
        Ma = mates[am].shape[1]
        gpos = int(ba[r, 0, 1] & 0xFFFFFFFF)
        wb = gpos - 15 if gpos > 15 else 0; we = min(wb + 31 + Ma, G)
        ok, sc_, src_, snk_, cig_, _ = orc.banded_gotoh_traceback(31, oracle.SEMI_GLOBAL, osc, oriented(mates[am][r], a_rc), text[wb:we])
        k = am + 1
        assert sc_ == ba[r, 0, 0] and int(tb["score%d" % k][r]) == sc_, (mode, r, "anchor score")
        assert int(tb["begin%d" % k][r]) == wb + src_[0] and int(tb["rc%d" % k][r]) == a_rc, (mode, r, "anchor begin")
        n = int(tb["cigar_lens%d" % k][r])
        assert n == len(cig_) and np.array_equal(tb["cigars%d" % k][r, :n].cpu().numpy().view(np.uint16), cig_), (mode, r, "anchor cigar")
        ob = int(bo[r, 0, 1] & 0xFFFFFFFF); osx = int(bo[r, 0, 2] & 0xFFFFFFFF)
        ok, sc_, src_, snk_, cig_ = orc.full_gotoh_traceback(oracle.SEMI_GLOBAL, osc, oriented(mates[1 - am][r], o_rc), text[ob:ob + osx])
        k = 2 - am
        assert sc_ == bo[r, 0, 0], (mode, r, "opposite score")
        assert int(tb["begin%d" % k][r]) == ob + src_[0] and int(tb["rc%d" % k][r]) == o_rc, (mode, r, "opposite begin")
        n = int(tb["cigar_lens%d" % k][r])
        assert n == len(cig_) and np.array_equal(tb["cigars%d" % k][r, :n].cpu().numpy().view(np.uint16), cig_), (mode, r, "opposite cigar")
        checked += 1
    assert checked > 20
    fmi.close()


@pytest.mark.parametrize("max_hits", [100, 7])
def test_approximate_seed_mapper_equals_the_oracle(amd, orc, max_hits):
    """seed_mapper<APPROX_MAPPING>: four one-mismatch searches per seed over the forward index and the index of the reversed text; the
    deques (hit for hit, in heap order, with the reference's flags) and the reseeding decisions equal the oracle's restatement"""
    import torch
    rng = np.random.default_rng(55)
    G = 120_000
    text = rng.integers(0, 4, G, dtype=np.uint8)
    text[3000:3400] = np.tile(text[3000:3010], 40)                 # a tandem repeat: many one-mismatch neighbours
    hidx, ridx = orc.build_index(text), orc.build_index(text[::-1].copy())
    fmi = amd.FMIndex.build(orc.pack2(text), G, kmer_len=0, sa_int=16)
    rfmi = amd.FMIndex.build(orc.pack2(text[::-1].copy()), G, kmer_len=0, sa_int=16)
    R, M, L, S = 600, 100, 20, 13
    starts = rng.integers(0, G - M, R); starts[:40] = rng.integers(2990, 3300, 40)
    reads = np.stack([text[s:s + M] for s in starts]).copy()
    mut = rng.random(reads.shape) < 0.03
    reads[mut] = (reads[mut] + 1 + rng.integers(0, 3, int(mut.sum()))) % 4
    reads[rng.random(reads.shape) < 0.003] = 4
    rcm = rng.random(R) < 0.5
    reads[rcm] = np.where(reads[rcm][:, ::-1] < 4, 3 - reads[rcm][:, ::-1], 4)
    stored = np.ascontiguousarray(reads[:, ::-1])
    spr = (M - L) // S + 1
    sp = amd.SeedHitsParams(spr, S, L, M, max_hits=max_hits, rep_seeds=50)
    cap = amd.seed_hits_approx_capacity(sp)
    deques = torch.zeros((R, cap, 2), dtype=torch.int32, device="cuda:0")
    sizes = torch.zeros(R, dtype=torch.int32, device="cuda:0"); reseed = torch.zeros(R, dtype=torch.uint8, device="cuda:0")
    for bits, packed in ((4, orc.pack4(stored.reshape(-1))), (8, stored.reshape(-1))):
        dt = torch.from_numpy(packed.view(np.int32) if bits == 4 else packed).cuda()
        amd.seed_hits_map_approx(fmi, rfmi, dt, bits, sp, R, deques, sizes, reseed)
        d, n, rs = amd.u32(deques), sizes.cpu().numpy(), reseed.cpu().numpy()
        total = 0
        for r in range(R):
            want, want_rs = orc.map_approx_read(hidx, ridx, stored[r], np.arange(spr) * S, L, max_hits, 50)
            assert n[r] == len(want) and np.array_equal(d[r, :n[r]], want) and bool(rs[r]) == want_rs, (bits, r)
            total += len(want)
        assert total > 5 * R                                        # plenty of hits, and with max_hits = 7 the cap bites
    fmi.close(); rfmi.close()
