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

from . import (ALN_RAGGED_READS, FM_COMPLEMENT, FM_SCAN_FORWARD, LOCAL, READ_COMPLEMENT, READ_REVERSE, SCORE_MIN, SEMI_GLOBAL, AlignmentBatch,
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
        self.max_seed_hits = max_seed_hits          # None: every SA row of every seed range is extended (fmmap); k: only
                                                    # the first k rows of a seed's SA range (a deterministic stand-in for
                                                    # nvBowtie's max_hits cap, which guards against repeat seeds)
        self.direct = True                          # use match_direct when the index holds the full SA and the text
        self.fused_seed_pass = True                 # ... and run match + scan + locate + diagonal keys + adjacent dedupe of
                                                    # every single-hit seed in ONE kernel (nvbio_fm_match_seed_diagonals)
        self.mapq = False                           # also keep nvBowtie's second-best alignment per read and compute the
                                                    # mapping quality (score_reduce + BowtieMapq2); results go to `extras`
        self.mapq_version = 2
        self.merge_strands = True                   # one-call seed pass: extend the candidates of both strands in ONE batch (half the
                                                    # launches, better-filled kernels) instead of strand by strand
        self.algo_flags = 0                         # nvbio_alignment_batch::algo_flags of the extension (ALN_*: A/B of the exact shortcuts)
        self.defer_heavy = True                     # two-strand seed pass: the searches the table cannot answer (k-mers with more than 8
                                                    # occurrences) run as a dense launch of their own behind the pass (FM_DEFER_HEAVY)

    @classmethod
    def end_to_end(cls, constant_quality=True, **kw):
        """nvBowtie's default (end-to-end) mode: GotohAligner<SEMI_GLOBAL> (scoring.h:212) with the default
        SmithWatermanScoringScheme (scoring_inl.h:99-114): match 0, mismatch 2..6 by quality, gaps 5+3 / 3.
        constant_quality: reads carry no qualities and every base counts as q >= 40 (mismatch -6)."""
        scheme = GotohScheme(0, 6, 6, -8, -3, -8, -3) if constant_quality else GotohScheme(0, 2, 6, -8, -3, -8, -3)
        p = cls(aln_type=SEMI_GLOBAL, scheme=scheme, **kw)
        p.e2e = True
        return p

    def interval_for(self, read_len):
        return self.seed_interval or int(1 + 1.15 * math.sqrt(read_len))

    def min_score_for(self, read_len):
        if self.min_score is not None:
            return self.min_score
        if getattr(self, "e2e", False):                # linear -0.6 - 0.6 L (scoring_inl.h:99-114, MinScoreFunc scoring.h:117-129)
            return int(np.float32(-0.6) + np.float32(-0.6) * np.float32(read_len))
        return int(np.float32(0.0) + np.float32(10.0) * np.float32(math.log(np.float32(read_len))))

    def interval_table(self, max_len):
        """seed interval of a read of every length 0..max_len: nvBowtie's seed_freq( read_len ) = int32( 1 + 1.15 sqrtf( len ) )
        (SimpleFunc, params.h:87-100; bowtie2_cuda_driver.cu:111-113), evaluated in float32 as the reference does"""
        if self.seed_interval:
            return np.full(max_len + 1, self.seed_interval, dtype=np.int32)
        x = np.arange(max_len + 1, dtype=np.float32)
        return np.maximum((np.float32(1.0) + np.float32(1.15) * np.sqrt(x)).astype(np.int32), 1)

    def min_score_table(self, max_len):
        """min_score_for() of every read length 0..max_len (ragged batches: the threshold is the read's own)"""
        return np.array([self.min_score_for(max(l, 1)) for l in range(max_len + 1)], dtype=np.int32)


class ReadBatch:
    """io::SequenceData<DNA_N>-shaped batch in HBM: 4-bit big-endian packed symbols, one quality byte per symbol (optional).
    Uniform length: read r = symbols [r * read_len, (r + 1) * read_len).  Ragged (offsets: int32 tensor [n_reads + 1], the
    sequence_index): read r = [offsets[r], offsets[r+1]); read_len is then the LONGEST read's length."""

    def __init__(self, reads4, n_reads, read_len, quals=None, offsets=None):
        self.reads4, self.n, self.read_len, self.quals = reads4, int(n_reads), int(read_len), quals
        self.offsets = offsets


def _batch_geometry(torch, fmi, reads, params):
    """seed enumeration of a read batch: the seed set (infixes [r*M + j*S, +L) of the read stream, enumerated inside the kernels:
    uniform_seeds_functor semantics, no offset array is materialised), the reads' offsets and, for a ragged batch, every read's own
    seed interval (seed_freq( read_len ), mapping_inl.h:507-529) and score threshold -- the tables are evaluated on the host in float32
    as the reference evaluates them, per length, and gathered per read on the device"""
    dev = fmi.device
    R, M, L = reads.n, reads.read_len, params.seed_len
    S_int = params.interval_for(M)
    spr = (M - L) // S_int + 1                                   # seeds per read and strand
    ragged = reads.offsets is not None
    intervals = min_scores = None
    if ragged:
        cache = getattr(reads, "_ragged", None)
        if cache is None or cache[0] is not params:
            lens = (reads.offsets[1:] - reads.offsets[:-1]).to(torch.int64)
            itab = params.interval_table(M)
            it = torch.from_numpy(itab).to(dev)
            mt = torch.from_numpy(params.min_score_table(M)).to(dev)
            spr_max = max(((l - L) // int(itab[l]) + 1) if l >= L else 0 for l in range(M + 1))
            cache = reads._ragged = (params, it[lens].contiguous(), mt[lens].contiguous(), int(spr_max))
        _, intervals, min_scores, spr = cache
        qs = PackedStringSet(reads.reads4, 4, R * spr, offsets=reads.offsets, fixed_len=L, stride=0, device=dev, seeds_per_string=spr,
                             seed_intervals=intervals)
        read_off = reads.offsets
    else:
        qs = PackedStringSet(reads.reads4, 4, R * spr, fixed_len=L, stride=M, device=dev, seeds_per_string=spr,
                             seed_interval=S_int)
        read_off = getattr(reads, "_read_off", None)
        if read_off is None:
            read_off = reads._read_off = torch.arange(R + 1, device=dev, dtype=torch.int32) * M
    return dict(S_int=S_int, spr=spr, ragged=ragged, intervals=intervals, min_scores=min_scores, qs=qs, read_off=read_off)


def seed_pass_begin(fmi, reads, params, slot=0, timers=None, stream=None, after=None):
    """Enqueue the two-strand seed pass of a batch and the copy of its counts to pinned host memory; returns the handle
    seed_and_extend( ..., pre = handle ) continues from.  A caller that streams batches enqueues batch i+1's seed pass BEFORE batch i's
    extension: the host then reads i+1's counts while the GPU extends batch i, and the one host synchronisation of a step -- the sizes
    of the extension's launches -- never leaves the GPU idle (bench.py does this).  slot: which of the handle's buffer sets to use
    (two consecutive batches need different ones).
    stream: a side stream to enqueue the pass on, so that it runs BESIDE the previous batch's extension (the seed pass waits for HBM lines, the
    extension for the VALU: the two share a CU well); after: an event of the caller's stream behind the last kernel that read this slot's
    buffers (the extension of two batches ago) -- the pass waits for it on the device."""
    import torch
    if stream is not None:
        if after is not None:
            stream.wait_event(after)
        with torch.cuda.stream(stream):
            return seed_pass_begin(fmi, reads, params, slot, timers)
    geo = _batch_geometry(torch, fmi, reads, params)
    bufs = getattr(fmi, "_seed_bufs2", None)
    if bufs is None:
        bufs = fmi._seed_bufs2 = {}
    b = bufs.setdefault(slot, {})
    ev_t = None
    if timers is not None:
        a, ev_t = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        timers.setdefault("match_both", []).append((a, ev_t))
        a.record()
    # seeds that end on 2..4 rows (short repeats) leave all their keys at once: on unique-ish genomes the residual lists stay
    # empty and the scan + locate path with its host round trips is not entered at all
    fmi.match_seed_diagonals_both(geo["qs"], reads.read_len, b, inline_hits=min(4, params.max_seed_hits or 4), defer_heavy=params.defer_heavy,
                                  grid_blocks=int(getattr(params, "seed_grid_blocks", 0) or 0))
    if ev_t is not None:
        ev_t.record()
    if "host" not in b:
        b["host"] = torch.empty(4, dtype=torch.int32, pin_memory=True)
    b["host"].copy_(b["counts"][:4], non_blocking=True)
    ev = torch.cuda.Event()
    ev.record()
    return dict(b=b, ev=ev, reads=reads)


SCORE_BIAS = 1 << 20


def pack_best_key(torch, scores, rc, pos):
    """order-independent selection key: highest score, then reverse strand, then the larger end
    position (fmmap only reduces the score per read, examples/fmmap/fmmap.cu:367-376; the locus
    tie-break is this pipeline's own and does not depend on the order candidates are listed in)"""
    s = torch.clamp(scores.to(torch.int64) + SCORE_BIAS, min=0)
    return (s << 34) | (rc.to(torch.int64) << 33) | pos.to(torch.int64)


def seed_and_extend(fmi, genome2, genome_len, reads, params, timers=None, return_windows=False, extras=None, pre=None):
    """returns (best_score[int32 R], best_pos[int64 R] (text position of the alignment's end, or -1),
    best_rc[uint8 R], n_candidates).  timers: optional dict name -> list of (start, end) events.
    return_windows: also return best_wb[int64 R], the window begin of each read's best candidate (-1 if
    none; the largest one if several candidates tie on the whole selection key) -- what traceback_best needs --
    and best_g[int64 R], that candidate's locus (hit.loc: the diagonal clamped at the genome start), the
    anchor position of paired-end opposite-mate windows.
    params.mapq: `extras` (a dict) receives "mapq" (uint8 [R]), "second_score" (int32 [R], SCORE_MIN where a read has no
    second alignment) and "second" (the second-best selection keys): nvBowtie's score_reduce bookkeeping (reduce_inl.h:65-140,
    over the candidates in descending key order) and BowtieMapq2 (mapq.h)."""
    import torch
    from . import best_candidate_reduce, best_candidate_unpack, best_candidate_windows, diagonals_to_windows, mapq, second_candidate_reduce
    dev = fmi.device
    R, M, L = reads.n, reads.read_len, params.seed_len
    geo = _batch_geometry(torch, fmi, reads, params)
    S_int, spr, ragged, intervals, min_scores, qs, read_off = (geo[k] for k in ("S_int", "spr", "ragged", "intervals", "min_scores", "qs", "read_off"))

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

    ext_flags = (params.algo_flags or 0) | (ALN_RAGGED_READS if ragged else 0)
    aligner = GotohAligner(params.aln_type, params.scheme)
    top = torch.zeros((R,), dtype=torch.int64, device=dev)      # best selection key per read (0: no candidate)

    def extend(keys, tag):
        """candidate windows (genome_infixes, fmmap.cu:169-196; window rule of score_inl.h:100-106), the banded Gotoh of
        every one of them, and the per-read reduction of the selection keys into `top` (one atomic max per candidate)"""
        e = tick("windows" + tag)
        rid, flags, wb, we = diagonals_to_windows(keys, params.band, M, genome_len, read_offsets=read_off if ragged else None)
        tock(e)
        e = tick("extend" + tag)
        batch = AlignmentBatch(reads.reads4, 4, read_off, genome2, 2, wb, we, quals=reads.quals, read_id=rid,
                               flags=flags, device=dev, max_read_len=M, algo_flags=ext_flags or None)
        scores, sinks = BatchedBandedAlignmentScore(params.band, aligner).enact(batch)
        tock(e)
        e = tick("reduce")
        best_candidate_reduce(keys, scores, sinks, wb, top)
        tock(e)
        if params.mapq or return_windows:
            scored.append((keys, scores, sinks, wb))
        return None

    use_direct = params.direct and fmi.supports_direct()
    results, n_cand, scored = [], 0, []
    fused = use_direct and params.fused_seed_pass

    def residual_keys(ranges, ids, strand):
        """seeds that ended on several SA rows (repeats): the ordinary scan + locate path -> their diagonal keys"""
        if params.max_seed_hits is not None and params.max_seed_hits <= 64 and use_direct and ids.numel() * params.max_seed_hits < 2 ** 31 \
                and getattr(params, "one_call_residuals", True):
            # under a seed-hit cap: one call (sort by seed id, capped expansion at a fixed stride, neighbour-duplicate removal, select)
            if strand:
                ids = ids ^ -2 ** 31
            rk, nk = fmi.residual_diagonals(ranges.contiguous(), ids.contiguous(), params.max_seed_hits, spr, S_int, L, M,
                                            read_offsets=read_off if ragged else None, seed_intervals=intervals)
            return rk[:int(nk.item())]
        if params.max_seed_hits is not None:
            x = ranges[:, 0].to(torch.int64) & 0xFFFFFFFF
            y = ranges[:, 1].to(torch.int64) & 0xFFFFFFFF
            ycap = torch.minimum(y, x + (params.max_seed_hits - 1))
            ranges = torch.stack([ranges[:, 0], torch.where(ycap >= 2 ** 31, ycap - 2 ** 32, ycap).to(torch.int32)], dim=1).contiguous()
        flt = FMIndexFilter()
        n_hits = flt.rank_ranges(fmi, ranges.contiguous(), None)
        rkeys = flt.locate_diagonals(0, n_hits, spr, S_int, L, M, strand, query_ids=ids.contiguous(),
                                     read_offsets=read_off if ragged else None, seed_intervals=intervals)
        rkeys = torch.unique_consecutive(rkeys)
        if rkeys.numel() > 2 * R:
            rkeys = torch.unique(rkeys)
        return rkeys

    ck = fmi.canonical_kmer if fused else 0
    both = fused and ck and ck <= L <= ck + 7 and spr <= 64 and getattr(params, "two_strand_pass", True)
    if ragged and fused and not both:
        fused = False                   # (the per-strand one-call pass serves uniform batches only: a ragged batch takes the separate operators)
    if both:
        # 2.-4. for BOTH strands in one kernel over the canonical table (a k-mer and its reverse complement share an entry): one table
        #    gather per seed window instead of one per window and strand
        if pre is None:
            pre = seed_pass_begin(fmi, reads, params, 0, timers)
        assert pre["reads"] is reads
        b = pre["b"]
        pre["ev"].synchronize()
        n_keys, n_rf, n_rr = int(b["host"][0]), int(b["host"][1]), int(b["host"][2])
        parts = [b["keys"][:n_keys]]
        if n_rf or n_rr:
            # both strands' residual seeds through ONE scan + locate (bit 31 of a seed id flips the strand)
            e = tick("locate")
            parts.append(residual_keys(torch.cat([b["ranges"][:n_rf], b["ranges"][qs.n:qs.n + n_rr]]),
                                       torch.cat([b["ids"][:n_rf], b["ids"][qs.n:qs.n + n_rr] | -2 ** 31]), 0))
            tock(e)
        keys = parts[0] if len(parts) == 1 else torch.cat(parts)
        n_cand += keys.numel()
        if keys.numel():
            results.append(extend(keys, ""))
    elif fused:
        # 2.-4. in one kernel per strand: every seed that ends on one SA row leaves it as a deduplicated diagonal key; the few
        #    that end on several rows (repeats) come back as a residual list and take the scan + locate path.  Both strands'
        #    seed passes are enqueued first; each one's two counts travel to pinned host memory behind it, so that the host
        #    learns the forward strand's candidate count while the reverse strand's pass runs and the GPU never waits for it.
        bufs = getattr(fmi, "_seed_bufs", None)
        if bufs is None:
            bufs = fmi._seed_bufs = ({}, {})
        pending = []
        for strand, flags in ((0, 0), (1, FM_SCAN_FORWARD | FM_COMPLEMENT)):
            b = bufs[strand]
            e = tick("match_fw" if strand == 0 else "match_rc")
            fmi.match_seed_diagonals(qs, flags, M, strand, b)
            tock(e)
            if "host" not in b:
                b["host"] = torch.empty(2, dtype=torch.int32, pin_memory=True)
            b["host"].copy_(b["counts"][:2], non_blocking=True)
            ev = torch.cuda.Event()
            ev.record()
            pending.append((strand, b, ev))
        merged = []
        for strand, b, ev in pending:
            ev.synchronize()
            n_keys, n_res = int(b["host"][0]), int(b["host"][1])
            keys = b["keys"][:n_keys]
            if n_res:
                e = tick("locate")
                keys = torch.cat([keys, residual_keys(b["ranges"][:n_res], b["ids"][:n_res], strand)])
                tock(e)
            if keys.numel() == 0:
                continue
            n_cand += keys.numel()
            if params.merge_strands:
                merged.append(keys)
            else:
                results.append(extend(keys, "_rc" if strand else "_fw"))
        if merged:
            results.append(extend(merged[0] if len(merged) == 1 else torch.cat(merged), ""))
    for strand, flags in (() if fused else ((0, 0), (1, FM_SCAN_FORWARD | FM_COMPLEMENT))):
        # 2. exact-match every seed: SA ranges + inclusive scan of their sizes
        #    (FMIndexFilter::rank = match + scan; the two halves are called separately so that the
        #    match kernel can be timed on its own)
        flt = FMIndexFilter()
        e = tick("match_fw" if strand == 0 else "match_rc")
        if use_direct:
            # seeds whose range collapses to one row finish on the text and come back as positions
            ranges, direct = fmi.match_direct(qs, flags)
        else:
            ranges, direct = fmi.match(qs, flags), None
        tock(e)
        if params.max_seed_hits is not None:
            # keep the first max_seed_hits rows of every range (empty ranges, x > y, stay empty; direct ones hold 1 hit)
            x = ranges[:, 0].to(torch.int64) & 0xFFFFFFFF
            y = ranges[:, 1].to(torch.int64) & 0xFFFFFFFF
            ycap = torch.minimum(y, x + (params.max_seed_hits - 1))
            ranges = torch.stack([ranges[:, 0], torch.where(ycap >= 2 ** 31, ycap - 2 ** 32, ycap).to(torch.int32)], dim=1).contiguous()
        e = tick("scan")
        n_hits = flt.rank_ranges(fmi, ranges, direct)
        tock(e)
        if n_hits == 0:
            continue
        # 3. + 4. SA rows -> text positions -> diagonal keys (hit_to_diagonal, examples/fmmap/fmmap.cu:92-117) in one
        #    pass over the hits; consecutive seeds of a read that agree on the diagonal collapse to one candidate (hits
        #    arrive in seed order, so an adjacent compare removes nearly all duplicates without a sort; a survivor only
        #    costs a repeated extension)
        e = tick("locate")
        keys = flt.locate_diagonals(0, n_hits, spr, S_int, L, M, strand, read_offsets=read_off if ragged else None, seed_intervals=intervals)
        tock(e)
        e = tick("diagonals")
        keys = torch.unique_consecutive(keys)
        if keys.numel() > 2 * R:
            # repeats: the seeds of a read list the same loci over and over, interleaved, so the adjacent compare misses
            # them; a sort-based unique costs far less than extending every copy (not taken on unique-ish genomes)
            keys = torch.unique(keys)
        tock(e)
        n_cand += keys.numel()
        # (running this strand's VALU-bound extension on a second stream beside the other strand's
        # memory-bound seed pass was measured: the two kernels serialise, 42-44 ms vs 36 ms per step)
        results.append(extend(keys, "_rc" if strand else "_fw"))

    # 5. best candidate per read (already reduced into `top` by extend): unpack the keys
    e = tick("unpack")
    best_score, best_pos, best_rc = best_candidate_unpack(top)
    tock(e)
    if params.mapq:
        # 6. second-best per read (needs the final best of both strands) and the mapping quality
        e = tick("mapq")
        second = torch.zeros((R,), dtype=torch.int64, device=dev)
        min_score = params.min_score_for(M)
        for keys, scores, sinks, wb in scored:
            second_candidate_reduce(keys, scores, sinks, wb, top, M // 2, min_score - 1, second,
                                    read_offsets=read_off if ragged else None, min_scores=min_scores)
        match = int(params.scheme.c.match)
        q, second_score = mapq(top, second, match * M, min_score, match == 0, params.mapq_version,
                               read_offsets=read_off if ragged else None, min_scores=min_scores, match=match)
        tock(e)
        if extras is not None:
            extras.update(mapq=q, second_score=second_score, second=second)
    if n_cand == 0:
        if return_windows:
            none = torch.full((R,), -1, dtype=torch.int64, device=dev)
            return best_score, best_pos, best_rc, 0, none, none.clone()
        return best_score, best_pos, best_rc, 0

    if extras is not None:
        extras["best_keys"] = top
        if ragged:
            extras["min_scores"] = min_scores
    if return_windows:
        # the window begin and the locus of every read's best candidate: one pass over the candidates (those whose selection key is
        # their read's final best)
        best_wb = torch.full((R,), -1, dtype=torch.int64, device=dev)
        best_g = torch.full((R,), -1, dtype=torch.int64, device=dev)
        for keys, scores, sinks, wb in scored:
            best_candidate_windows(keys, scores, sinks, wb, top, best_wb, best_g)
        return best_score, best_pos, best_rc, int(n_cand), best_wb, best_g
    return best_score, best_pos, best_rc, int(n_cand)


def traceback_best_all(genome2, genome_len, reads, params, best_keys, best_wb, cigar_stride=32, timers=None):
    """nvBowtie's banded_traceback_best (traceback_inl.h:191-247) over ALL reads of the batch, job r = read r, the batch built on the
    device from the per-read best keys (extras["best_keys"] of seed_and_extend) and best_wb: no compaction, no host round trip.
    The scoring pass's score and sink are handed over (end-to-end; a LOCAL alignment is re-scored by the traceback).  Returns
    (scores [R], align_pos [R] int64 = text position where the alignment starts, sources, sinks, cigars [R, cigar_stride], cigar_lens
    [R]); a read that did not align has cigar_len 0 and sink (-1, -1)."""
    import torch
    from . import BatchedBandedAlignmentTraceback, traceback_best_batch
    dev = best_keys.device
    R, M = reads.n, reads.read_len
    ev = None
    if timers is not None:
        a, ev = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        timers.setdefault("traceback", []).append((a, ev))
        a.record()
    if reads.offsets is not None:
        read_off = reads.offsets
        flags, wb, we, scores, sinks = traceback_best_batch(best_keys, best_wb, M, params.band, genome_len, 0, read_offsets=read_off,
                                                            min_scores=reads._ragged[2])
    else:
        flags, wb, we, scores, sinks = traceback_best_batch(best_keys, best_wb, M, params.band, genome_len, params.min_score_for(M))
        read_off = getattr(reads, "_read_off", None)
        if read_off is None:
            read_off = reads._read_off = torch.arange(R + 1, device=dev, dtype=torch.int32) * M
    batch = AlignmentBatch(reads.reads4, 4, read_off, genome2, 2, wb, we, quals=reads.quals, flags=flags, device=dev, max_read_len=M)
    known = dict(scores=scores, sinks=sinks) if params.aln_type != LOCAL else {}
    op = BatchedBandedAlignmentTraceback(params.band, GotohAligner(params.aln_type, params.scheme))
    # the direction vectors of the gapped alignments (a minority) go to the persistent scratch: a multi-GiB allocation
    # per call stalls the step for 0.1-0.2 s whenever the HIP pool has handed the memory back (measured)
    from . import _scratch
    temp = _scratch(dev, op.min_temp_storage(batch) // 4, cap=6 << 30)
    sc, src, snk, cig, ln = op.enact(batch, cigar_stride=cigar_stride, temp=temp, **known)
    pos = (wb.to(torch.int64) & 0xFFFFFFFF) + (src[:, 0].to(torch.int64) & 0xFFFFFFFF)
    if ev is not None:
        ev.record()
    return sc, pos, src, snk, cig, ln


def traceback_best(genome2, genome_len, reads, params, best_score, best_rc, best_wb, cigar_stride=32, timers=None,
                   best_pos=None):
    """nvBowtie's banded_traceback_best (traceback_inl.h:191-247) for the pipeline's best alignment per read:
    re-align every aligned read inside its best candidate's window with the traceback kernel.
    Returns (read ids [A], scores, align_pos [A] int64 = text position where the alignment starts,
    sources, sinks, cigars [A, cigar_stride] (io::Cigar elements, backtracking order), cigar_lens)."""
    import torch
    from . import BatchedBandedAlignmentTraceback
    dev = best_score.device
    R, M = reads.n, reads.read_len
    ids = torch.nonzero((best_wb >= 0) & (best_score >= params.min_score_for(M))).view(-1)      # aligned reads only
    wb = best_wb[ids]
    we = torch.clamp(wb + params.band + M, max=genome_len)
    flags = (best_rc[ids].to(torch.uint8) * (READ_REVERSE | READ_COMPLEMENT)).to(torch.uint8)
    read_off = torch.arange(R + 1, device=dev, dtype=torch.int32) * M

    def i32(t):
        return torch.where(t >= 2 ** 31, t - 2 ** 32, t).to(torch.int32)

    batch = AlignmentBatch(reads.reads4, 4, read_off, genome2, 2, i32(wb), i32(we), quals=reads.quals,
                           read_id=ids.to(torch.int32), flags=flags, device=dev, max_read_len=M)
    ev = None
    if timers is not None:
        a, ev = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        timers.setdefault("traceback", []).append((a, ev))
        a.record()
    known = {}
    if best_pos is not None and params.aln_type != LOCAL:
        # the scoring pass already ran on exactly these windows: hand its score and sink over (an end-to-end
        # alignment ends in the last pattern row, so the sink is (end position - window begin, read length))
        sx = (best_pos[ids] - wb).to(torch.int32)
        known = dict(scores=best_score[ids].contiguous(), sinks=torch.stack([sx, torch.full_like(sx, M)], dim=1).contiguous())
    from . import _scratch
    op = BatchedBandedAlignmentTraceback(params.band, GotohAligner(params.aln_type, params.scheme))
    sc, src, snk, cig, ln = op.enact(batch, cigar_stride=cigar_stride, temp=_scratch(dev, op.min_temp_storage(batch) // 4, cap=6 << 30), **known)
    if ev is not None:
        ev.record()
    pos = wb + (src[:, 0].to(torch.int64) & 0xFFFFFFFF)
    return ids, sc, pos, src, snk, cig, ln



class PairedEndParams:
    """nvBowtie's paired-end defaults (bowtie2_cuda_driver.cu:116-120): FR policy, fragments up to 500, mates may overlap"""

    def __init__(self, policy=None, min_frag_len=0, max_frag_len=500, overlap=True):
        from . import PE_POLICY_FR
        self.policy = PE_POLICY_FR if policy is None else policy
        self.min_frag_len, self.max_frag_len, self.overlap = min_frag_len, max_frag_len, overlap


def paired_end(fmi, genome2, genome_len, mates1, mates2, params, pe=None, timers=None, cigar_stride=0):
    """A paired-end composition of the path's operators in nvBowtie's shape: each mate in turn is the anchor --
    seed-and-extend as a single end, best candidate per read -- and the other mate is scored by full-matrix DP inside the
    window the fragment-length constraints allow around the anchor (BestOppositeScoreStream, score_inl.h:283-456), with
    min_score = the opposite mate's own worst admissible score.  The pair with the higher anchor + opposite score wins
    (ties: mate 1 as anchor).  Returns a dict of per-pair tensors: anchor (0/1, -1 if no pair), score1/2, pos1/2 (end
    positions), rc1/2.  (nvBowtie additionally iterates over several anchor candidates and tightens min_score with the
    pairs found so far; that policy is not part of this path.)
    cigar_stride > 0: also trace the chosen pair back -- the anchor mate through the banded traceback in its candidate
    window, the opposite mate through the full-matrix traceback in its window (nvBowtie's banded_traceback_best /
    traceback_best) -- adding begin1/2 (text position where each mate's alignment starts), cigars1/2 [R, cigar_stride]
    (io::Cigar runs, backtracking order) and cigar_lens1/2 (0 for unpaired reads)."""
    import torch
    from . import (BatchedAlignmentScore, BatchedAlignmentTraceback, BatchedBandedAlignmentTraceback, max_text_gaps,
                   opposite_mate_windows)
    pe = pe or PairedEndParams()
    dev = fmi.device
    R = mates1.n
    assert mates2.n == R
    worst = -(1 << 30)
    out = {}
    cand = []
    trace_state = []
    for anchor, (a, o) in enumerate(((mates1, mates2), (mates2, mates1))):
        tag = "_a%d" % anchor
        sub = None if timers is None else {}
        bs, bp, brc, nc, bwb, bg = seed_and_extend(fmi, genome2, genome_len, a, params, sub, return_windows=True)
        if timers is not None:
            for k, v in sub.items():
                timers.setdefault(k, []).extend(v)
        ids = torch.nonzero((bg >= 0) & (bs >= params.min_score_for(a.read_len))).view(-1)
        o_min = params.min_score_for(o.read_len)
        gaps = max_text_gaps(params.scheme, o_min, o.read_len)
        g32 = torch.where(bg[ids] >= 2 ** 31, bg[ids] - 2 ** 32, bg[ids]).to(torch.int32)
        wb, we, flags, valid = opposite_mate_windows(g32, brc[ids].contiguous(), a.read_len, o.read_len + gaps, anchor, genome_len,
                                                     pe.policy, pe.min_frag_len, pe.max_frag_len, pe.overlap)
        keep = torch.nonzero(valid).view(-1)
        ids, wb, we, flags = ids[keep], wb[keep].contiguous(), we[keep].contiguous(), flags[keep].contiguous()
        read_off = torch.arange(R + 1, device=dev, dtype=torch.int32) * o.read_len
        batch = AlignmentBatch(o.reads4, 4, read_off, genome2, 2, wb, we, quals=o.quals, read_id=ids.to(torch.int32), flags=flags,
                               device=dev, max_read_len=o.read_len)
        ms = torch.full((ids.numel(),), o_min, dtype=torch.int32, device=dev)
        ev = None
        if timers is not None:
            a_ev, ev = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            timers.setdefault("opposite" + tag, []).append((a_ev, ev)); a_ev.record()
        osc, osk = BatchedAlignmentScore(GotohAligner(params.aln_type, params.scheme), text_blocking=False).enact(
            batch, o.read_len, pe.max_frag_len, min_scores=ms)
        if ev is not None:
            ev.record()
        ok = osc >= o_min                                            # hit.opposite_score = score >= min_score ? score : worst
        o_score = torch.full((R,), worst, dtype=torch.int64, device=dev)
        o_pos = torch.full((R,), -1, dtype=torch.int64, device=dev)
        o_rc = torch.zeros((R,), dtype=torch.uint8, device=dev)
        sel = ids[ok]
        o_score[sel] = osc[ok].to(torch.int64)
        o_pos[sel] = (wb[ok].to(torch.int64) & 0xFFFFFFFF) + (osk[ok][:, 0].to(torch.int64) & 0xFFFFFFFF)
        o_rc[sel] = (flags[ok] != 0).to(torch.uint8)
        cand.append((bs.to(torch.int64), bp, brc, o_score, o_pos, o_rc))
        if cigar_stride:
            o_wb = torch.full((R,), -1, dtype=torch.int64, device=dev); o_we = o_wb.clone()
            o_wb[sel] = wb[ok].to(torch.int64) & 0xFFFFFFFF; o_we[sel] = we[ok].to(torch.int64) & 0xFFFFFFFF
            o_sx = torch.zeros((R,), dtype=torch.int32, device=dev); o_sx[sel] = osk[ok][:, 0]
            trace_state.append((bwb, o_wb, o_we, o_sx))
    (s1a, p1a, r1a, s2a, p2a, r2a), (s2b, p2b, r2b, s1b, p1b, r1b) = cand
    pair_a = torch.where(s2a > worst, s1a + s2a, torch.full_like(s1a, worst))
    pair_b = torch.where(s1b > worst, s2b + s1b, torch.full_like(s1b, worst))
    use_b = pair_b > pair_a
    paired = (pair_a > worst) | (pair_b > worst)
    out["anchor"] = torch.where(paired, use_b.to(torch.int64), torch.full_like(pair_a, -1))
    out["score1"] = torch.where(use_b, s1b, s1a); out["pos1"] = torch.where(use_b, p1b, p1a); out["rc1"] = torch.where(use_b, r1b, r1a)
    out["score2"] = torch.where(use_b, s2b, s2a); out["pos2"] = torch.where(use_b, p2b, p2a); out["rc2"] = torch.where(use_b, r2b, r2a)
    out["pair_score"] = torch.maximum(pair_a, pair_b)
    if not cigar_stride:
        return out

    # ---- traceback of the chosen pairs ------------------------------------------------------------------------------
    def i32(t):
        return torch.where(t >= 2 ** 31, t - 2 ** 32, t).to(torch.int32)

    for m in (1, 2):
        out["begin%d" % m] = torch.full((R,), -1, dtype=torch.int64, device=dev)
        out["cigars%d" % m] = torch.zeros((R, cigar_stride), dtype=torch.int16, device=dev)
        out["cigar_lens%d" % m] = torch.zeros((R,), dtype=torch.int32, device=dev)
    aligner = GotohAligner(params.aln_type, params.scheme)
    for anchor, (a, o) in enumerate(((mates1, mates2), (mates2, mates1))):
        bwb, o_wb, o_we, o_sx = trace_state[anchor]
        ids = torch.nonzero(out["anchor"] == anchor).view(-1)
        if ids.numel() == 0:
            continue
        am, om = (1, 2) if anchor == 0 else (2, 1)
        a_rc = out["rc%d" % am][ids]; o_rc = out["rc%d" % om][ids]
        rid32 = ids.to(torch.int32)
        e = None
        if timers is not None:
            s_ev, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            timers.setdefault("traceback_a%d" % anchor, []).append((s_ev, e)); s_ev.record()
        # anchor mate: banded, in the window of its best candidate
        wb = bwb[ids]; we = torch.clamp(wb + params.band + a.read_len, max=genome_len)
        a_off = torch.arange(R + 1, device=dev, dtype=torch.int32) * a.read_len
        batch = AlignmentBatch(a.reads4, 4, a_off, genome2, 2, i32(wb), i32(we), quals=a.quals, read_id=rid32,
                               flags=(a_rc.to(torch.uint8) * (READ_REVERSE | READ_COMPLEMENT)).to(torch.uint8), device=dev,
                               max_read_len=a.read_len)
        # (direction vectors to the persistent scratch: a multi-GiB allocation inside the call stalls for ~0.1 s
        # whenever the HIP pool has handed the memory back)
        from . import _scratch
        tb_op = BatchedBandedAlignmentTraceback(params.band, aligner)
        _, src, _, cig, ln = tb_op.enact(batch, cigar_stride=cigar_stride, temp=_scratch(dev, tb_op.min_temp_storage(batch) // 4, cap=6 << 30))
        out["begin%d" % am][ids] = wb + (src[:, 0].to(torch.int64) & 0xFFFFFFFF)
        out["cigars%d" % am][ids] = cig; out["cigar_lens%d" % am][ids] = ln
        # opposite mate: full matrix, in the opposite-mate window; its score and sink are known from the scoring pass
        wb = o_wb[ids]; we = o_we[ids]
        o_off = torch.arange(R + 1, device=dev, dtype=torch.int32) * o.read_len
        batch = AlignmentBatch(o.reads4, 4, o_off, genome2, 2, i32(wb), i32(we), quals=o.quals, read_id=rid32,
                               flags=(o_rc.to(torch.uint8) * (READ_REVERSE | READ_COMPLEMENT)).to(torch.uint8), device=dev,
                               max_read_len=o.read_len)
        o_min = params.min_score_for(o.read_len)
        ms = torch.full((ids.numel(),), o_min, dtype=torch.int32, device=dev)
        known = {}
        if params.aln_type != LOCAL:
            sx = o_sx[ids]
            known = dict(scores=out["score%d" % om][ids].to(torch.int32).contiguous(),
                         sinks=torch.stack([sx, torch.full_like(sx, o.read_len)], dim=1).contiguous())
        _, src, _, cig, ln = BatchedAlignmentTraceback(aligner).enact(batch, o.read_len, pe.max_frag_len, min_scores=ms,
                                                                     cigar_stride=cigar_stride, **known)
        out["begin%d" % om][ids] = wb + (src[:, 0].to(torch.int64) & 0xFFFFFFFF)
        out["cigars%d" % om][ids] = cig; out["cigar_lens%d" % om][ids] = ln
        if e is not None:
            e.record()
    return out



class NvBowtieParams:
    """the parameters of nvBowtie's best-approx pipeline that decide WHICH loci get extended (bowtie2_cuda_driver.cu:86-141 defaults)"""

    def __init__(self, seed_len=22, seed_freq=None, max_hits=100, rep_seeds=1000, max_effort=15, max_effort_init=15, min_ext=30,
                 max_ext=400, max_reseed=2, band=31, top_seed=0):
        self.seed_len, self.seed_freq, self.max_hits, self.rep_seeds = seed_len, seed_freq, max_hits, rep_seeds
        self.max_effort, self.max_effort_init, self.min_ext, self.max_ext = max_effort, max(max_effort_init, max_effort), min_ext, max(max_ext, max_effort)
        self.max_reseed, self.band, self.top_seed = max_reseed, band, top_seed


def nvbowtie_best_approx(fmi, genome2, genome_len, stored_reads, params, nvb=None, stats=None):
    """nvBowtie's best-approx single-end loop with nvBowtie's own choices (Aligner::best_approx, aligner_best_approx.h:39-207,363-667):
    for every seeding pass (reads that ask for reseeding go round again with their seeds shifted), map the exact seeds of both strands
    into per-read hit deques capped at max_hits (the smallest SA ranges survive), then repeat: select the next SA row of every active
    read's smallest range -> locate it -> band-DP its window (BestScoreStream) -> fold the score into the read's best / second best
    in arrival order, counting failed extensions, until a read's hits or its effort run out.  One hit per read and pass (the
    reference switches to several once fewer than half a batch of reads are active: an optimisation that changes no rule but
    the order effort runs out in).  Every data-parallel step is a kernel behind the C ABI; this function is the host loop.
    stored_reads: ReadBatch of reads stored REVERSED, as nvBowtie loads them (io::REVERSE, nvBowtie.cpp:322).
    Returns dict(best_score, best_loc, best_rc, second_score, second_loc, second_rc) (loc = hit.loc, the diagonal's locus; -1 = none),
    n_extensions, passes."""
    import torch
    from . import (FM_COMPLEMENT, FM_SCAN_FORWARD, AlignmentBatch, BatchedBandedAlignmentScore, GotohAligner, HitQueues, PackedStringSet,
                   SeedHitsParams, score_reduce_effort, score_stream_flatten, score_stream_output, seed_hits_loc, seed_hits_map, seed_hits_select)
    nvb = nvb or NvBowtieParams()
    dev = fmi.device
    R, M = stored_reads.n, stored_reads.read_len
    L = min(nvb.seed_len, M)
    S = nvb.seed_freq or params.interval_for(M)
    retry_stride = S // (nvb.max_reseed + 1)
    worst = params.min_score_for(M)                                   # init_alignments( reads, threshold_score, ... ) (aligner_best_approx.h:77)
    best = torch.empty((R, 4), dtype=torch.int32, device=dev)
    best[:, 0] = worst; best[:, 2] = worst; best[:, 1] = -1; best[:, 3] = -1
    best_rc = torch.zeros(R, dtype=torch.uint8, device=dev)
    trys = torch.empty(R, dtype=torch.int32, device=dev)
    read_index = torch.arange(R + 1, device=dev, dtype=torch.int32) * M
    aligner = GotohAligner(params.aln_type, params.scheme)
    queue = torch.arange(R, device=dev, dtype=torch.int32)           # seed_queues: the reads of this seeding pass
    n_extensions = passes = 0
    count = torch.zeros(1, dtype=torch.int32, device=dev)
    for seeding_pass in range(nvb.max_reseed + 1):
        nq = queue.numel()
        if nq == 0:
            break
        first = seeding_pass * retry_stride
        spr = (M - L - first) // S + 1 if M >= L + first else 0
        if spr <= 0:
            break
        sp = SeedHitsParams(spr, S, L, M, first_offset=first, max_hits=nvb.max_hits, rep_seeds=nvb.rep_seeds, max_effort=nvb.max_effort,
                            min_ext=nvb.min_ext, max_ext=nvb.max_ext)
        cap = sp.capacity()
        # the two match_range calls of the exact mapper over the seeds of the queued reads
        offs = (queue.to(torch.int64) * M + first).to(torch.int32).contiguous()
        qs = PackedStringSet(stored_reads.reads4, 4, nq * spr, offsets=offs, fixed_len=L, stride=M, device=dev, seeds_per_string=spr, seed_interval=S)
        fw = fmi.match(qs, FM_SCAN_FORWARD)
        rc = fmi.match(qs, FM_COMPLEMENT)
        deques = torch.zeros((R, cap, 2), dtype=torch.int32, device=dev)
        sizes = torch.zeros(R, dtype=torch.int32, device=dev)
        reseed = torch.zeros(R, dtype=torch.uint8, device=dev)
        seed_hits_map(fw, rc, sp, nq, deques, sizes, reseed, read_queue=queue)
        # the extension loop (best_approx_score)
        trys.fill_(nvb.max_effort_init)                               # select_init
        active = (queue | (nvb.top_seed << 31)).contiguous()
        n_ext = 0
        while active.numel() and n_ext < nvb.max_ext:
            na = active.numel()
            hits = HitQueues(torch.empty(na, dtype=torch.int32, device=dev), torch.empty(na, dtype=torch.int32, device=dev),
                             torch.empty(na, dtype=torch.int32, device=dev), device=dev)
            active_out = torch.empty(na, dtype=torch.int32, device=dev)
            seed_hits_select(active, trys, sp, deques, sizes, hits, active_out, count)
            nh = int(count.item())
            if nh == 0:
                break
            active = active_out[:nh].contiguous()
            hits.n = nh
            pos = fmi.locate(hits.loc[:nh].contiguous())
            seed_hits_loc(pos, hits)
            rid, flags, wb, we = score_stream_flatten(hits, read_index, nvb.band, genome_len, reads_reversed=True)
            batch = AlignmentBatch(stored_reads.reads4, 4, read_index, genome2, 2, wb, we, quals=stored_reads.quals, read_id=rid, flags=flags,
                                   device=dev, max_read_len=M)
            scores, sinks = BatchedBandedAlignmentScore(nvb.band, aligner).enact(batch)
            score_stream_output(hits, scores, sinks, wb)
            score_reduce_effort(active, hits, M, n_ext, sp, best, best_rc, trys, sizes)
            n_ext += 1
            n_extensions += nh
            passes += 1
        queue = queue[reseed[queue.to(torch.int64)] != 0].contiguous()       # the reads that asked for reseeding go round again
    if stats is not None:
        stats.update(n_extensions=n_extensions, passes=passes)
    b = best.to(torch.int64)
    loc = lambda c: torch.where(b[:, c] == -1, b[:, c], b[:, c] & 0xFFFFFFFF)
    return dict(best_score=best[:, 0].clone(), best_loc=loc(1), best_rc=(best_rc & 1), second_score=best[:, 2].clone(), second_loc=loc(3),
                second_rc=((best_rc >> 1) & 1), n_extensions=n_extensions, passes=passes)


# ---- the same loop as a C++ host loop over the C ABI (host/nvbio_amd/best_approx.hpp behind lib/libnvbio_amd_host.so) --------------------------
_HOST_LIB = None


def _host_lib():
    import ctypes
    import os
    global _HOST_LIB
    if _HOST_LIB is None:
        from . import lib as _core
        _core()                                                       # libnvbio_amd.so (and torch's HIP runtime) first
        path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib", "libnvbio_amd_host.so")
        if not os.path.exists(path):
            raise RuntimeError("%s is missing: build it with __graft_entry__.build()" % path)
        L = ctypes.CDLL(path)
        L.nvbio_host_last_error.restype = ctypes.c_char_p
        _HOST_LIB = L
    return _HOST_LIB


def nvbowtie_best_approx_host(fmi, genome2, genome_len, stored_reads, params, nvb=None, batch_size=0, multi_hit=True):
    """nvbowtie_best_approx as the C++ host loop (nvbio_host_best_approx): every data-parallel step behind the C ABI, the loop in C++, two
    counters read per extension pass through pinned memory, queues allocated once -- and the reference's several-hits-per-read phase
    (aligner_best_approx.h:487-510; batch_size = its BATCH_SIZE, 0 = the number of reads).  Same return value as nvbowtie_best_approx."""
    import ctypes
    import torch
    from . import FMIndex, _ptr, _stream_ptr
    nvb = nvb or NvBowtieParams()
    dev = fmi.device
    R, M = stored_reads.n, stored_reads.read_len

    class _P(ctypes.Structure):
        _fields_ = [(k, ctypes.c_uint32) for k in ("seed_len", "seed_freq", "max_hits", "rep_seeds", "max_effort", "max_effort_init", "min_ext", "max_ext",
                                                  "max_reseed", "band", "top_seed", "batch_size", "multi_hit")]

    class _S(ctypes.Structure):
        _fields_ = [("n_extensions", ctypes.c_uint64), ("passes", ctypes.c_uint32), ("multi_passes", ctypes.c_uint32), ("seeding_passes", ctypes.c_uint32),
                    ("pad", ctypes.c_uint32)]

    p = _P(nvb.seed_len, nvb.seed_freq or params.interval_for(M), nvb.max_hits, nvb.rep_seeds, nvb.max_effort, nvb.max_effort_init, nvb.min_ext, nvb.max_ext,
           nvb.max_reseed, nvb.band, nvb.top_seed, int(batch_size), 1 if multi_hit else 0)
    st = _S()
    best = torch.empty((R, 4), dtype=torch.int32, device=dev)
    best_rc = torch.zeros(R, dtype=torch.uint8, device=dev)
    rc = _host_lib().nvbio_host_best_approx(
        ctypes.c_int(FMIndex._dev_index(dev)), fmi._h, _ptr(genome2), ctypes.c_uint32(genome_len), _ptr(stored_reads.reads4), _ptr(stored_reads.quals),
        ctypes.c_uint32(R), ctypes.c_uint32(M), ctypes.c_int(int(params.aln_type)), ctypes.byref(params.scheme.c), ctypes.c_int32(params.min_score_for(M)),
        ctypes.byref(p), _ptr(best), _ptr(best_rc), _stream_ptr(dev), ctypes.byref(st))
    if rc != 0:
        raise RuntimeError(_host_lib().nvbio_host_last_error().decode())
    b = best.to(torch.int64)
    loc = lambda c: torch.where(b[:, c] == -1, b[:, c], b[:, c] & 0xFFFFFFFF)
    return dict(best_score=best[:, 0].clone(), best_loc=loc(1), best_rc=(best_rc & 1), second_score=best[:, 2].clone(), second_loc=loc(3),
                second_rc=((best_rc >> 1) & 1), n_extensions=int(st.n_extensions), passes=int(st.passes), multi_passes=int(st.multi_passes),
                seeding_passes=int(st.seeding_passes))


def nvbowtie_best_approx_paired_host(fmi, genome2, genome_len, stored_mates1, stored_mates2, params, nvb=None, pe=None, unpaired=True, batch_size=0, multi_hit=True):
    """nvBowtie's PAIRED-END best-approx loop (Aligner::best_approx, aligner_best_approx_paired.h:84-200,590-1000) as the C++ host loop
    nvbio_host_best_approx_paired: for anchor = mate 1, then mate 2, the single-end loop over the anchor's seed hits with every selected hit scored as
    a pair -- the anchor against a threshold that tightens with the pairs found so far (compute_target_score), the opposite mate by full-matrix DP in
    its fragment window for the hits whose anchor passed, score_reduce_paired keeping the best two pairs (or per-mate bests while unpaired).
    -> dict( best_a, best_o [R, 2, 4] int32 = { score, position, sink offset, rc | mate << 1 | paired << 2 } x { best, second }, counters )."""
    import ctypes
    import torch
    from . import FMIndex, _ptr, _stream_ptr
    nvb = nvb or NvBowtieParams()
    pe = pe or PairedEndParams()
    dev = fmi.device
    R, M1, M2 = stored_mates1.n, stored_mates1.read_len, stored_mates2.read_len
    if stored_mates2.n != R:
        raise ValueError("the two mate batches must hold the same number of reads")

    class _P(ctypes.Structure):
        _fields_ = [(k, ctypes.c_uint32) for k in ("seed_len", "seed_freq", "max_hits", "rep_seeds", "max_effort", "max_effort_init", "min_ext", "max_ext",
                                                  "max_reseed", "band", "top_seed", "batch_size", "multi_hit")]

    class _PE(ctypes.Structure):
        _fields_ = [(k, ctypes.c_uint32) for k in ("policy", "min_frag_len", "max_frag_len", "overlap", "unpaired")]

    class _S(ctypes.Structure):
        _fields_ = [("n_extensions", ctypes.c_uint64), ("n_opposite", ctypes.c_uint64), ("passes", ctypes.c_uint32), ("multi_passes", ctypes.c_uint32)]

    p = _P(nvb.seed_len, nvb.seed_freq or 0, nvb.max_hits, nvb.rep_seeds, nvb.max_effort, nvb.max_effort_init, nvb.min_ext, nvb.max_ext, nvb.max_reseed, nvb.band,
           nvb.top_seed, int(batch_size), 1 if multi_hit else 0)
    q = _PE(int(pe.policy), int(pe.min_frag_len), int(pe.max_frag_len), 1 if pe.overlap else 0, 1 if unpaired else 0)
    st = _S()
    best_a = torch.empty((R, 2, 4), dtype=torch.int32, device=dev)
    best_o = torch.empty((R, 2, 4), dtype=torch.int32, device=dev)
    rc = _host_lib().nvbio_host_best_approx_paired(
        ctypes.c_int(FMIndex._dev_index(dev)), fmi._h, _ptr(genome2), ctypes.c_uint32(genome_len), _ptr(stored_mates1.reads4), _ptr(stored_mates2.reads4),
        _ptr(stored_mates1.quals), _ptr(stored_mates2.quals), ctypes.c_uint32(R), ctypes.c_uint32(M1), ctypes.c_uint32(M2), ctypes.c_int(int(params.aln_type)),
        ctypes.byref(params.scheme.c), ctypes.c_int32(params.min_score_for(M1)), ctypes.c_int32(params.min_score_for(M2)), ctypes.byref(p), ctypes.byref(q),
        _ptr(best_a), _ptr(best_o), _stream_ptr(dev), ctypes.byref(st))
    if rc != 0:
        raise RuntimeError(_host_lib().nvbio_host_last_error().decode())
    return dict(best_a=best_a, best_o=best_o, n_extensions=int(st.n_extensions), n_opposite=int(st.n_opposite), passes=int(st.passes),
                multi_passes=int(st.multi_passes))


def nvbowtie_paired_traceback(genome2, genome_len, stored_mates1, stored_mates2, params, loop, nvb=None, cigar_stride=32):
    """The traceback stage behind nvBowtie's paired loop (Aligner::best_approx: banded_traceback_best for the anchor mate, traceback_best for the
    opposite mate, traceback_inl.h:191-275): for every pair the loop found (best_a[:, 0] paired), the anchor mate is traced back through the banded
    traceback in the window BestAnchorScoreStream scored it in, the opposite mate through the full-matrix traceback from its window's begin to the
    column its alignment ends in -- scores and sinks are the loop's, so neither scoring pass is repeated.
    loop: the dict of nvbowtie_best_approx_paired_host.  -> dict( paired [R] bool; per mate m = 1, 2: begin{m} (text position where the alignment
    starts, -1 unpaired), rc{m}, score{m}, cigars{m} [R, cigar_stride] (io::Cigar runs, backtracking order), cigar_lens{m} )."""
    import torch
    from . import BatchedAlignmentTraceback, BatchedBandedAlignmentTraceback
    nvb = nvb or NvBowtieParams()
    dev = loop["best_a"].device
    R = stored_mates1.n
    mates = (stored_mates1, stored_mates2)
    a1 = loop["best_a"][:, 0].to(torch.int64); o1 = loop["best_o"][:, 0].to(torch.int64)
    paired = (((a1[:, 3] >> 2) & 1) == 1) & (a1[:, 1] != -1)
    out = {"paired": paired}
    for m in (1, 2):
        out["begin%d" % m] = torch.full((R,), -1, dtype=torch.int64, device=dev)
        out["rc%d" % m] = torch.zeros((R,), dtype=torch.uint8, device=dev)
        out["score%d" % m] = torch.zeros((R,), dtype=torch.int32, device=dev)
        out["cigars%d" % m] = torch.zeros((R, cigar_stride), dtype=torch.int16, device=dev)
        out["cigar_lens%d" % m] = torch.zeros((R,), dtype=torch.int32, device=dev)
    aligner = GotohAligner(params.aln_type, params.scheme)

    def i32(t):
        return torch.where(t >= 2 ** 31, t - 2 ** 32, t).to(torch.int32)

    def flags_of(rc):                                                  # reads stored reversed: forward = READ_REVERSE, reverse-complemented = READ_COMPLEMENT
        return torch.where(rc.bool(), torch.full_like(rc, READ_COMPLEMENT), torch.full_like(rc, READ_REVERSE)).to(torch.uint8)

    for am in (0, 1):                                                  # pairs whose anchor is mate am + 1
        ids = torch.nonzero(paired & (((a1[:, 3] >> 1) & 1) == am)).view(-1)
        if ids.numel() == 0:
            continue
        a, o = mates[am], mates[1 - am]
        rid32 = ids.to(torch.int32)
        # anchor: the band window of pe_anchor_flatten around hit.loc; sink.x = hit.sink - window begin
        g = a1[ids, 1] & 0xFFFFFFFF
        a_rc = (a1[ids, 3] & 1).to(torch.uint8)
        wb = torch.where(g > nvb.band // 2, g - nvb.band // 2, torch.zeros_like(g))
        we = torch.clamp(wb + nvb.band + a.read_len, max=genome_len)
        sx = g + (a1[ids, 2] & 0xFFFFFFFF) - wb
        a_off = torch.arange(R + 1, device=dev, dtype=torch.int32) * a.read_len
        batch = AlignmentBatch(a.reads4, 4, a_off, genome2, 2, i32(wb), i32(we), quals=a.quals, read_id=rid32, flags=flags_of(a_rc), device=dev, max_read_len=a.read_len)
        sinks = torch.stack([i32(sx), torch.full_like(i32(sx), a.read_len)], dim=1).contiguous()
        _, src, _, cig, ln = BatchedBandedAlignmentTraceback(nvb.band, aligner).enact(batch, cigar_stride=cigar_stride, scores=a1[ids, 0].to(torch.int32).contiguous(), sinks=sinks)
        k = am + 1
        out["begin%d" % k][ids] = wb + (src[:, 0].to(torch.int64) & 0xFFFFFFFF)
        out["rc%d" % k][ids] = a_rc; out["score%d" % k][ids] = a1[ids, 0].to(torch.int32)
        out["cigars%d" % k][ids] = cig; out["cigar_lens%d" % k][ids] = ln
        # opposite mate: from its window's begin (alignment.pos) to the column it ends in (pos + sink)
        ob = o1[ids, 1] & 0xFFFFFFFF
        osx = o1[ids, 2] & 0xFFFFFFFF
        o_rc = (o1[ids, 3] & 1).to(torch.uint8)
        o_off = torch.arange(R + 1, device=dev, dtype=torch.int32) * o.read_len
        batch = AlignmentBatch(o.reads4, 4, o_off, genome2, 2, i32(ob), i32(ob + osx), quals=o.quals, read_id=rid32, flags=flags_of(o_rc), device=dev, max_read_len=o.read_len)
        sinks = torch.stack([i32(osx), torch.full_like(i32(osx), o.read_len)], dim=1).contiguous()
        max_text = int(osx.max().item()) if ids.numel() else 1
        _, src, _, cig, ln = BatchedAlignmentTraceback(aligner).enact(batch, o.read_len, max(max_text, 1), cigar_stride=cigar_stride,
                                                                     scores=o1[ids, 0].to(torch.int32).contiguous(), sinks=sinks)
        k = 2 - am
        out["begin%d" % k][ids] = ob + (src[:, 0].to(torch.int64) & 0xFFFFFFFF)
        out["rc%d" % k][ids] = o_rc; out["score%d" % k][ids] = o1[ids, 0].to(torch.int32)
        out["cigars%d" % k][ids] = cig; out["cigar_lens%d" % k][ids] = ln
    return out
