"""GPU parity of the staged scheduler's banded scoring (nvbio_banded_gotoh_score_staged =
BatchedBandedAlignmentScore<BAND, stream, DeviceStagedThreadScheduler>, batched_banded_inl.h:165-236: 32-row windows with the
min_score exit) against the reference's own outputs (tests/golden/staged_golden.npz) and, on an nvBowtie-shaped packed batch,
against the oracle."""
import numpy as np
import pytest

import oracle
from util import mutate_reads

pytestmark = pytest.mark.gpu


def _staged(amd, band, typ, sv, batch, min_scores):
    op = amd.BatchedBandedAlignmentScore(band, amd.make_gotoh_aligner(typ, amd.GotohScheme(*[int(x) for x in sv])),
                                         scheduler=amd.DEVICE_STAGED_THREAD_SCHEDULER)
    return op.enact(batch, min_scores=min_scores)


def test_staged_golden_bytes(amd, staged_golden):
    import torch
    g = staged_golden
    noq = np.array([g["quals"][g["pat_off"][k]] == 255 for k in range(len(g["band"]))])
    seen = early = 0
    for band in (3, 7, 15, 31):
        for typ in range(3):
            for si in range(len(g["schemes"])):
                for hq in (False, True):
                    sel = np.nonzero((g["band"] == band) & (g["typ"] == typ) & (g["scheme"] == si) & (noq != hq))[0].astype(np.uint32)
                    if len(sel) == 0:
                        continue
                    batch = amd.AlignmentBatch(g["pats"], 8, g["pat_off"], g["txts"], 8, g["txt_off"][sel], g["txt_off"][sel + 1],
                                               quals=g["quals"] if hq else None, read_id=sel)
                    ms = torch.tensor(g["min_score"][sel].astype(np.int32), device="cuda:0")
                    sc, sk = _staged(amd, band, typ, g["schemes"][si], batch, ms)
                    want = g["out"][sel]
                    assert np.array_equal(sc.cpu().numpy().astype(np.int64), want[:, 1]), (band, typ, si, hq)
                    assert np.array_equal(amd.u32(sk).astype(np.int64), want[:, 2:4] & 0xFFFFFFFF), (band, typ, si, hq)
                    seen += len(sel); early += int((want[:, 0] == 0).sum())
    assert seen == len(g["band"]) and early > 500


@pytest.mark.parametrize("typ,sv", [(oracle.LOCAL, (2, 2, 6, -8, -3, -8, -3)), (oracle.SEMI_GLOBAL, (0, 2, 6, -8, -3, -8, -3))])
def test_staged_packed_batch_equals_the_oracle(amd, orc, typ, sv):
    """nvBowtie's shape: 4-bit reads stored reversed (fw hits read backwards, rc hits complemented), windows on a 2-bit genome,
    min_score = what BestScoreStream hands over (max(second best, score limit), score_inl.h:112-113): one value for all jobs and
    one per job.  With min_score out of reach the call must equal the plain one; with a real limit, reads drawn from elsewhere stop
    at their first window."""
    import torch
    rng = np.random.default_rng(41)
    G, R, M, J = 200000, 3000, 150, 9000
    text = rng.integers(0, 4, G, dtype=np.uint8)
    starts = rng.integers(20, G - M - 40, R)
    reads = mutate_reads(rng, text, starts, M)
    roffs = (np.arange(R + 1) * M).astype(np.uint32)
    quals = rng.integers(0, 64, R * M, dtype=np.uint8)
    rid = rng.integers(0, R, J).astype(np.uint32)
    flags = np.where(rng.random(J) < 0.5, 1, 0).astype(np.uint8)              # NVBIO_READ_REVERSE on half of the jobs
    g_pos = starts[rid].astype(np.int64)
    g_pos[::3] = rng.integers(20, G - M - 40, len(g_pos[::3]))                # a third of the jobs look at an unrelated locus
    wb = (g_pos - 15).astype(np.uint32); we = (wb + M + 31).astype(np.uint32)
    stored = reads.copy()
    batch = amd.AlignmentBatch(orc.pack4(stored.reshape(-1)), 4, roffs, orc.pack2(text), 2, wb, we, quals=quals, read_id=rid,
                               flags=flags, max_read_len=M)
    plain_sc, plain_sk = amd.batch_banded_alignment_score(31, amd.make_gotoh_aligner(typ, amd.GotohScheme(*sv)), batch)
    sc, sk = _staged(amd, 31, typ, sv, batch, None)                           # Field_traits<int32>::min(): never stops
    assert torch.equal(sc, plain_sc) and torch.equal(sk, plain_sk)
    # LOCAL: the exit test is max(band) < min_score + rows_left * match AS WRITTEN (gotoh_banded_inl.h:619-621): with a match bonus the
    # limit has to lie 2 * rows_left BELOW what the band holds for a job to go on -- -180 lets the true loci through
    limit = -180 if typ == oracle.LOCAL else -60
    per_job = np.where(rng.random(J) < 0.5, limit, limit - 40).astype(np.int32)
    for ms_arg, ms_np in ((limit, np.full(J, limit, np.int32)), (torch.tensor(per_job, device="cuda:0"), per_job)):
        sc, sk = _staged(amd, 31, typ, sv, batch, ms_arg)
        sc, sk = sc.cpu().numpy(), amd.u32(sk)
        stopped = 0
        for j in range(0, J, 2):
            pat = reads[rid[j]][::-1] if flags[j] else reads[rid[j]]
            q = quals[rid[j] * M:(rid[j] + 1) * M]; q = q[::-1] if flags[j] else q
            ok, s_, k_ = orc.banded_gotoh_staged(31, typ, oracle.Scheme(*sv), pat, text[wb[j]:we[j]], int(ms_np[j]), q)
            assert sc[j] == s_ and tuple(int(v) for v in sk[j]) == (k_[0] & 0xFFFFFFFF, k_[1] & 0xFFFFFFFF), j
            stopped += ok == 0
        assert stopped > J // 8                                               # the unrelated loci stop early ...
        hit = np.arange(J) % 3 != 0
        fwd = hit & (flags == 0)
        assert (sc[fwd] == plain_sc.cpu().numpy()[fwd]).mean() > 0.9          # ... and the true ones run to the end


def test_staged_rejects_what_it_does_not_instantiate(amd):
    batch = amd.AlignmentBatch(np.zeros(4, np.uint8), 8, np.array([0, 4], np.uint32), np.zeros(8, np.int32), 2,
                               np.array([0], np.uint32), np.array([10], np.uint32))
    with pytest.raises(amd.NvbioError):
        _staged(amd, 31, 1, (2, 2, 6, -8, -3, -8, -3), batch, None)
    with pytest.raises(amd.NvbioError):
        amd.BatchedBandedAlignmentScore(31, amd.make_gotoh_aligner(1, amd.GotohScheme(2, 2, 6, -8, -3, -8, -3)), scheduler="WarpScheduler")
