"""GPU parity of nvBowtie's scoring stream handed over as data (nvbio_score_stream_flatten / nvbio_score_stream_output): what a
specialisation of aln::BatchedBandedAlignmentScore for bowtie2::cuda::BestScoreStream needs from the stream's pipeline object --
hit queues indexed through idx_queue, reads stored REVERSED, packed_seed::rc -- against the oracle's restatement of
BestScoreStream::init_context / load_strings / output (nvBowtie/bowtie2/cuda/score_inl.h:85-133, alignment_utils.h:277-302;
device-only sources: parity unpinned), and the whole path (flatten -> banded score -> output) against the oracle's DP run on
explicitly oriented reads."""
import numpy as np
import pytest

import oracle
from oracle import cpu_pipeline

pytestmark = pytest.mark.gpu


def _stream(rng, G, R, H, lens):
    """a scoring queue of H hits over R reads of the given lengths, in nvBowtie's layout"""
    read_index = np.zeros(R + 1, dtype=np.uint32); read_index[1:] = np.cumsum(lens)
    hit_read_id = rng.integers(0, R, H).astype(np.uint32)
    rc = rng.integers(0, 2, H).astype(np.uint32)
    pos_in_read = rng.integers(0, 4096, H).astype(np.uint32)
    hit_seed = (pos_in_read | (rng.integers(0, 2, H).astype(np.uint32) << 12) | (rc << 13) | (rng.integers(0, 2, H).astype(np.uint32) << 14)).astype(np.uint32)
    hit_loc = rng.integers(0, G, H).astype(np.uint32)
    hit_loc[:20] = rng.integers(0, 16, 20)                          # windows clipped at the genome start
    hit_loc[20:40] = G - rng.integers(1, 60, 20)                    # ... and at its end
    idx_queue = rng.permutation(H)[:H - H // 5].astype(np.uint32)   # the sorted scoring order touches a subset of the hits
    return read_index, hit_read_id, hit_seed, hit_loc, idx_queue, rc


@pytest.mark.parametrize("reads_reversed", [True, False])
def test_flatten_and_output_equal_the_restated_stream(amd, orc, reads_reversed):
    rng = np.random.default_rng(3)
    G, R, H = 500_000, 700, 5000
    lens = rng.integers(30, 160, R)
    read_index, rid, seed, loc, idxq, _ = _stream(rng, G, R, H, lens)
    for band in (3, 7, 15, 31):
        for q in (idxq, None):
            hq = amd.HitQueues(rid, seed, loc, idx_queue=q)
            got = amd.score_stream_flatten(hq, read_index, band, G, reads_reversed)
            want = cpu_pipeline.score_stream_flatten(q, rid, seed, loc, read_index, band, G, reads_reversed)
            import torch
            for g_, w_ in zip(got, want):
                assert np.array_equal(g_.cpu().numpy() if g_.dtype == torch.uint8 else amd.u32(g_), w_)
            n = len(want[0])
            scores = rng.integers(-200000, 300, n).astype(np.int32)
            sinks = np.stack([rng.integers(0, 200, n), rng.integers(0, 160, n)], axis=1).astype(np.uint32)
            amd.score_stream_output(hq, torch.from_numpy(scores).cuda(), torch.from_numpy(sinks.view(np.int32)).cuda(), got[2])
            ws, wk = cpu_pipeline.score_stream_output(q, H, scores, sinks, want[2])
            touched = np.zeros(H, dtype=bool); touched[q if q is not None else np.arange(H)] = True
            assert np.array_equal(hq.score.cpu().numpy()[touched], ws[touched])
            assert np.array_equal(amd.u32(hq.sink)[touched], wk[touched])
            assert (hq.score.cpu().numpy()[~touched] == 0).all()    # hits outside the queue are not written


def test_stream_to_scores_matches_the_reference_orientation(amd, orc):
    """reads stored reversed (io::REVERSE), hits on both strands: flatten -> nvbio_banded_gotoh_score -> output gives, for every
    hit, the score of the read as nvBowtie orients it (forward hit: the original read; rc hit: its reverse complement)
    against the window, computed by the oracle's DP on explicitly oriented symbols"""
    import torch
    rng = np.random.default_rng(8)
    G, R, H, M = 300_000, 400, 3000, 100
    text = rng.integers(0, 4, G, dtype=np.uint8)
    lens = np.full(R, M)
    read_index, rid, seed, loc, idxq, rc = _stream(rng, G, R, H, lens)
    # reads drawn near their hits' loci so that scores are informative
    reads = rng.integers(0, 4, (R, M), dtype=np.uint8)
    for h in range(0, H, 3):
        g = int(loc[h])
        if 20 < g < G - M - 40:
            r = text[g:g + M].copy()
            r[rng.integers(0, M, 3)] = rng.integers(0, 4, 3)
            reads[rid[h]] = (3 - r[::-1]) if rc[h] else r
    stored = reads[:, ::-1].copy()                                   # io::REVERSE
    band = 31
    hq = amd.HitQueues(rid, seed, loc, idx_queue=idxq)
    b_rid, b_flags, wb, we = amd.score_stream_flatten(hq, read_index, band, G, True)
    scheme = amd.GotohScheme(0, 6, 6, -8, -3, -8, -3)
    batch = amd.AlignmentBatch(orc.pack4(stored.reshape(-1)), 4, read_index, orc.pack2(text), 2, wb, we, read_id=b_rid, flags=b_flags, max_read_len=M)
    scores, sinks = amd.batch_banded_alignment_score(band, amd.make_gotoh_aligner(amd.SEMI_GLOBAL, scheme), batch)
    amd.score_stream_output(hq, scores, sinks, wb)
    hs, hk = hq.score.cpu().numpy(), amd.u32(hq.sink)
    osc = oracle.Scheme(0, 6, 6, -8, -3, -8, -3)
    wbn, wen = amd.u32(wb), amd.u32(we)
    for i in range(0, len(idxq), 7):
        h = int(idxq[i])
        pat = reads[rid[h]]
        pat = (3 - pat[::-1]) if rc[h] else pat                      # what load_strings hands the aligner
        _, score, sink = orc.banded_gotoh(band, oracle.SEMI_GLOBAL, osc, pat, text[wbn[i]:wen[i]])
        assert hs[h] == max(score, -65536), (i, h)
        if score > -(1 << 30):
            assert hk[h] == wbn[i] + sink[0], (i, h)
