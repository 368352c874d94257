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
