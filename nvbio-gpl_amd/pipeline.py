"""Seed-and-extend composition over the C-ABI kernels: the shape of the reference's smallest
complete caller (examples/fmmap/fmmap.cu:217-390: extract seeds -> FMIndexFilter::rank ->
FMIndexFilter::locate -> hit_to_diagonal -> banded alignment of the window around each
diagonal -> best score per read) with nvBowtie's seeding policy for exact seeds
(nvBowtie/bowtie2/cuda/mapping_inl.h:193-282: both strands; seed length 22, interval
int(1 + 1.15*sqrt(read_len)), bowtie2_cuda_driver.cu:86-141) and its scoring window
(BestScoreStream, nvBowtie/bowtie2/cuda/score_inl.h:85-115).

Every data-parallel stage runs in hand-written HIP behind include/nvbio_amd.h (match, scan,
expand + locate, banded Gotoh); torch is used for the index arithmetic between the stages
(hit -> diagonal, sort/unique of candidate loci, per-read arg-max), on the same stream.
This module never imports the oracle.
"""
import math

import numpy as np

from . import (FM_COMPLEMENT, FM_SCAN_FORWARD, LOCAL, READ_COMPLEMENT, READ_REVERSE, SCORE_MIN, AlignmentBatch,
               BatchedBandedAlignmentScore, FMIndexFilter, GotohAligner, GotohScheme, PackedStringSet)


class SeedExtendParams:
    def __init__(self, seed_len=22, seed_interval=None, band=31, aln_type=LOCAL, scheme=None, min_score=None,
                 max_seed_hits=None):
        self.seed_len = seed_len
        self.seed_interval = seed_interval          # None -> nvBowtie's S(1,1.15): int(1 + 1.15 sqrt(len))
        self.band = band
        self.aln_type = aln_type
        # nvBowtie local() scheme (scoring_inl.h:72-93): match 2, mismatch 2..6 by quality, gaps 5+3 / 3
        self.scheme = scheme or GotohScheme(2, 2, 6, -8, -3, -8, -3)
        self.min_score = min_score                  # None -> nvBowtie local(): int(0 + 10*ln(len)) (scoring.h:117-129)
        self.max_seed_hits = max_seed_hits          # None: every SA row of every seed range is extended (fmmap)

    def interval_for(self, read_len):
        return self.seed_interval or int(1 + 1.15 * math.sqrt(read_len))

    def min_score_for(self, read_len):
        if self.min_score is not None:
            return self.min_score
        return int(np.float32(0.0) + np.float32(10.0) * np.float32(math.log(np.float32(read_len))))


class ReadBatch:
    """io::SequenceData<DNA_N>-shaped batch in HBM: 4-bit big-endian packed symbols, uniform length."""

    def __init__(self, reads4, n_reads, read_len, quals=None):
        self.reads4, self.n, self.read_len, self.quals = reads4, int(n_reads), int(read_len), quals


def seed_and_extend(fmi, genome2, genome_len, reads, params, timers=None):
    """returns (best_score[int32 R], best_pos[int64 R] (text position of the alignment's end - or -1),
    best_rc[uint8 R], n_candidates).  timers: optional dict name -> (start_event, end_event) lists."""
    import torch
    dev = fmi.device
    R, M, L = reads.n, reads.read_len, params.seed_len
    S_int = params.interval_for(M)
    spr = (M - L) // S_int + 1                                   # seeds per read and strand

    def tick(name):
        if timers is None:
            return None
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        timers.setdefault(name, []).append((a, b))
        a.record()
        return b

    def tock(b):
        if b is not None:
            b.record()

    # 1. seeds: infixes [r*M + j*S, +L) of the read stream (uniform_seeds_functor semantics)
    e = tick("seed_offsets")
    read_off = torch.arange(R + 1, device=dev, dtype=torch.int64) * M
    offs = (read_off[:-1, None] + torch.arange(spr, device=dev, dtype=torch.int64)[None, :] * S_int).reshape(-1)
    offs32 = offs.to(torch.int32)
    qs = PackedStringSet(reads.reads4, 4, R * spr, offsets=offs32, fixed_len=L, device=dev)
    tock(e)

    cands = []
    n_hits_total = 0
    for strand, flags in ((0, 0), (1, FM_SCAN_FORWARD | FM_COMPLEMENT)):
        # 2. exact-match every seed: SA ranges + inclusive scan of their sizes
        flt = FMIndexFilter()
        e = tick("match_fw" if strand == 0 else "match_rc")
        n_hits = flt.rank(fmi, qs, flags)
        tock(e)
        if params.max_seed_hits is not None:
            raise NotImplementedError("max_seed_hits policy is not built yet")
        n_hits_total += n_hits
        if n_hits == 0:
            continue
        # 3. SA rows -> text positions, tagged with their seed
        e = tick("locate")
        hits = flt.locate(0, n_hits)
        tock(e)
        # 4. hit -> diagonal (examples/fmmap/fmmap.cu:92-117)
        e = tick("diagonals")
        pos = hits[:, 0].to(torch.int64) & 0xFFFFFFFF
        sid = hits[:, 1].to(torch.int64)
        rid = sid // spr
        p = (sid - rid * spr) * S_int
        if strand:
            p = M - p - L                                         # offset of the seed in the reverse-complemented read
        diag = pos - p + 1024
        cands.append((rid << 34) | (strand << 33) | diag)
        tock(e)

    best_score = torch.full((R,), SCORE_MIN, dtype=torch.int32, device=dev)
    best_pos = torch.full((R,), -1, dtype=torch.int64, device=dev)
    best_rc = torch.zeros((R,), dtype=torch.uint8, device=dev)
    if not cands:
        return best_score, best_pos, best_rc, 0

    # 5. candidate loci: unique (read, strand, diagonal), sorted by read
    e = tick("unique")
    keys = torch.unique(torch.cat(cands))
    C = keys.numel()
    rid = keys >> 34
    rc = (keys >> 33) & 1
    diag = (keys & ((1 << 33) - 1)) - 1024
    g_pos = torch.clamp(diag, min=0)
    half = params.band // 2
    wb = torch.where(g_pos > half, g_pos - half, torch.zeros_like(g_pos))          # score_inl.h:102-106
    we = torch.clamp(wb + params.band + M, max=genome_len)
    flags = (rc * (READ_REVERSE | READ_COMPLEMENT)).to(torch.uint8)
    tock(e)

    # 6. banded Gotoh of every candidate window
    e = tick("extend")
    batch = AlignmentBatch(reads.reads4, 4, read_off.to(torch.int32), genome2, 2, wb.to(torch.int32),
                           we.to(torch.int32), quals=reads.quals, read_id=rid.to(torch.int32), flags=flags, device=dev,
                           max_read_len=M)
    scores, sinks = BatchedBandedAlignmentScore(params.band, GotohAligner(params.aln_type, params.scheme)).enact(batch)
    tock(e)

    # 7. best candidate per read (ties: the last candidate in (strand, diagonal) order)
    e = tick("reduce")
    packed = (scores.to(torch.int64) << 32) | torch.arange(C, device=dev, dtype=torch.int64)
    top = torch.full((R,), -(1 << 62), dtype=torch.int64, device=dev)
    top.scatter_reduce_(0, rid, packed, "amax", include_self=True)
    has = top > -(1 << 62)
    ci = (top & 0xFFFFFFFF)[has]
    best_score[has] = scores[ci]
    best_pos[has] = wb[ci] + (sinks[ci, 0].to(torch.int64) & 0xFFFFFFFF)           # hit.sink = genome_begin + sink.x (score_inl.h:128-129)
    best_rc[has] = rc[ci].to(torch.uint8)
    tock(e)
    return best_score, best_pos, best_rc, int(C)
