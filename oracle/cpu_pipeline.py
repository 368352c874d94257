"""oracle/cpu_pipeline.py -- TEST INFRASTRUCTURE ONLY.

The reference-equivalent CPU path of the whole seed-and-extend composition, built from the
oracle's restatement of the reference's host functions (match / locate in an OpenMP parallel
for, as nvbio-test/fmindex_test.cu:375-407,879-890; BatchedBandedAlignmentScore<31,..,
HostThreadScheduler>, nvbio/alignment/batched_banded_inl.h:113-121).  nvBowtie itself has no CPU
path (its seeding / locate code is device-only), so "reference CPU aligned reads/s" is this
composition: 2 x 9 exact 22-mer seeds -> locate -> banded score per 150 bp read.

Used by tests/test_gpu_pipeline.py as the checker and by bench.py's cpu_baseline leg as the timed
baseline.  Never imported by the product.
"""
import math

import numpy as np

from . import LOCAL, SCORE_MIN, Oracle, Scheme


def revcomp(a):
    return (3 - a[..., ::-1]).astype(np.uint8) if (a <= 3).all() else np.where(a[..., ::-1] < 4, 3 - a[..., ::-1], a[..., ::-1]).astype(np.uint8)


def unpack2(words, begin, length):
    """2-bit big-endian packed words -> symbols [begin, begin+length) (vectorised)"""
    idx = begin + np.arange(length, dtype=np.int64)
    return ((words[idx >> 4] >> (30 - 2 * (idx & 15)).astype(np.uint32)) & 3).astype(np.uint8)


def seed_and_extend_ref(Rf, O, ridx, genome2, genome_len, reads, seed_len=22, seed_interval=None, band=31, aln_type=LOCAL,
                        scheme=None, timing=None):
    """The same composition run through the REFERENCE's own host code (oracle/_ref): match() and
    locate() in an OpenMP parallel for, banded_alignment_score<31> in an OpenMP parallel for.  The
    index arithmetic between them is numpy, as in seed_and_extend_cpu; unpacking the strings the
    reference's templates consume one symbol per byte is not counted in timing["ref_seconds"]."""
    import time
    scheme = scheme or Scheme(2, 2, 6, -8, -3, -8, -3)
    R, M = reads.shape
    L = seed_len
    S_int = seed_interval or int(1 + 1.15 * math.sqrt(M))
    spr = (M - L) // S_int + 1
    starts = np.arange(spr) * S_int
    t_ref = 0.0
    cands = []
    for strand in (0, 1):
        seeds = np.stack([reads[:, s:s + L] for s in starts], axis=1)
        if strand:
            seeds = np.where(seeds[..., ::-1] < 4, 3 - seeds[..., ::-1], seeds[..., ::-1]).astype(np.uint8)
        flat = np.ascontiguousarray(seeds.reshape(-1))
        offs = (np.arange(R * spr + 1) * L).astype(np.uint32)
        t0 = time.perf_counter()
        ranges = Rf.match_batch(ridx, flat, offs)
        t_ref += time.perf_counter() - t0
        sizes = np.where(ranges[:, 1] >= ranges[:, 0], ranges[:, 1].astype(np.int64) + 1 - ranges[:, 0], 0)
        total = int(sizes.sum())
        if total == 0:
            continue
        sid = np.repeat(np.arange(R * spr, dtype=np.int64), sizes)
        first = np.repeat(ranges[:, 0].astype(np.int64), sizes)
        local = np.arange(total, dtype=np.int64) - np.repeat(np.cumsum(sizes) - sizes, sizes)
        t0 = time.perf_counter()
        pos = Rf.locate_batch(ridx, (first + local).astype(np.uint32)).astype(np.int64)
        t_ref += time.perf_counter() - t0
        rid = sid // spr
        p = (sid - rid * spr) * S_int
        if strand:
            p = M - p - L
        cands.append((rid << 34) | (strand << 33) | (pos - p + 1024))
    best_score = np.full(R, SCORE_MIN, dtype=np.int32)
    best_pos = np.full(R, -1, dtype=np.int64)
    best_rc = np.zeros(R, dtype=np.uint8)
    if not cands:
        return best_score, best_pos, best_rc, 0
    keys = np.unique(np.concatenate(cands))
    C = len(keys)
    rid = keys >> 34
    rc = (keys >> 33) & 1
    diag = (keys & ((1 << 33) - 1)) - 1024
    g_pos = np.maximum(diag, 0)
    half = band // 2
    wb = np.where(g_pos > half, g_pos - half, 0)
    we = np.minimum(wb + band + M, genome_len)
    # strings one symbol per byte, as the reference's vector_view<const uint8*> strings
    pats = reads[rid]
    rcm = rc.astype(bool)
    pats[rcm] = np.where(pats[rcm][:, ::-1] < 4, 3 - pats[rcm][:, ::-1], pats[rcm][:, ::-1])
    tl = (we - wb).astype(np.int64)
    toff = np.zeros(C + 1, dtype=np.uint32); toff[1:] = np.cumsum(tl)
    tidx = np.repeat(wb, tl) + (np.arange(int(toff[-1]), dtype=np.int64) - np.repeat(toff[:-1].astype(np.int64), tl))
    txts = ((genome2[tidx >> 4] >> (30 - 2 * (tidx & 15)).astype(np.uint32)) & 3).astype(np.uint8)
    poff = (np.arange(C + 1) * M).astype(np.uint32)
    t0 = time.perf_counter()
    scores, sinks = Rf.banded_gotoh_batch(band, aln_type, scheme, np.ascontiguousarray(pats.reshape(-1)), poff, txts, toff)
    t_ref += time.perf_counter() - t0
    pos = wb.astype(np.int64) + sinks[:, 0].astype(np.int64)
    packed = (np.maximum(scores.astype(np.int64) + (1 << 20), 0) << 34) | (rc.astype(np.int64) << 33) | pos
    top = np.full(R, -1, dtype=np.int64)
    np.maximum.at(top, rid, packed)
    has = top >= 0
    sv = top >> 34
    best_score = np.where(has & (sv > 0), sv - (1 << 20), best_score).astype(np.int32)
    best_pos = np.where(has, top & ((1 << 33) - 1), best_pos)
    best_rc = np.where(has, (top >> 33) & 1, best_rc).astype(np.uint8)
    if timing is not None:
        timing["ref_seconds"] = t_ref
    return best_score, best_pos, best_rc, C


def seed_and_extend_cpu(O, hidx, genome_syms_or_packed, genome_len, reads, seed_len=22, seed_interval=None, band=31,
                        aln_type=LOCAL, scheme=None, quals=None, genome_is_packed=False, traceback_stride=0, traceback_min_score=None, want_loci=False,
                        max_seed_hits=None, second=None):
    """reads: uint8 [R, M] (values 0..4).  Returns (best_score, best_pos, best_rc, n_candidates); with
    second = dict(min_score, perfect_score, monotone, version) also a dict with nvBowtie's second-best alignment per read
    (score_reduce over the read's candidates in descending order of the selection key) and the mapping quality; with
    traceback_stride > 0 also a dict with the traceback of every aligned read's best candidate (the one with
    the largest window begin among candidates tying on the selection key)."""
    scheme = scheme or Scheme(2, 2, 6, -8, -3, -8, -3)
    R, M = reads.shape
    L = seed_len
    S_int = seed_interval or int(1 + 1.15 * math.sqrt(M))
    spr = (M - L) // S_int + 1
    starts = np.arange(spr) * S_int
    cands = []
    for strand in (0, 1):
        seeds = np.stack([reads[:, s:s + L] for s in starts], axis=1)            # [R, spr, L]
        if strand:
            seeds = np.where(seeds[..., ::-1] < 4, 3 - seeds[..., ::-1], seeds[..., ::-1]).astype(np.uint8)
        flat = np.ascontiguousarray(seeds.reshape(-1))
        offs = (np.arange(R * spr + 1) * L).astype(np.uint32)
        total, ranges, slots = O.filter_rank(hidx, flat, offs)
        if max_seed_hits is not None:                    # first max_seed_hits rows of every SA range
            x = ranges[:, 0].astype(np.int64); y = ranges[:, 1].astype(np.int64)
            ranges = np.stack([x, np.where(y >= x, np.minimum(y, x + max_seed_hits - 1), y)], axis=1).astype(np.uint32)
            sizes = np.where(ranges[:, 1].astype(np.int64) >= ranges[:, 0].astype(np.int64),
                             ranges[:, 1].astype(np.int64) + 1 - ranges[:, 0], 0)
            slots = np.cumsum(sizes).astype(np.uint64)
            total = int(sizes.sum())
        if total == 0:
            continue
        hits = O.filter_locate(hidx, ranges, slots, 0, total)
        pos = hits[:, 0].astype(np.int64)
        sid = hits[:, 1].astype(np.int64)
        rid = sid // spr
        p = (sid - rid * spr) * S_int
        if strand:
            p = M - p - L
        cands.append((rid << 34) | (strand << 33) | (pos - p + 1024))
    best_score = np.full(R, SCORE_MIN, dtype=np.int32)
    best_pos = np.full(R, -1, dtype=np.int64)
    best_rc = np.zeros(R, dtype=np.uint8)
    if not cands:
        if want_loci:
            return best_score, best_pos, best_rc, 0, np.full(R, -1, dtype=np.int64)
        return best_score, best_pos, best_rc, 0
    keys = np.unique(np.concatenate(cands))
    rid = keys >> 34
    rc = (keys >> 33) & 1
    diag = (keys & ((1 << 33) - 1)) - 1024
    g_pos = np.maximum(diag, 0)
    half = band // 2
    wb = np.where(g_pos > half, g_pos - half, 0)
    we = np.minimum(wb + band + M, genome_len)
    flags = (rc * 3).astype(np.uint8)
    reads4 = O.pack4(reads.reshape(-1))
    genome2 = genome_syms_or_packed if genome_is_packed else O.pack2(genome_syms_or_packed)
    roffs = (np.arange(R + 1) * M).astype(np.uint32)
    scores, sinks = O.banded_gotoh_packed_batch(band, aln_type, scheme, reads4, roffs, genome2, wb.astype(np.uint32),
                                                we.astype(np.uint32), read_id=rid.astype(np.uint32), flags=flags,
                                                quals=quals)
    # best candidate per read: highest score, then reverse strand, then the larger end position
    # (an order-independent rule; fmmap itself only reduces the score, examples/fmmap/fmmap.cu:367-376)
    pos = wb.astype(np.int64) + sinks[:, 0].astype(np.int64)
    packed = (np.maximum(scores.astype(np.int64) + (1 << 20), 0) << 34) | (rc.astype(np.int64) << 33) | pos
    top = np.full(R, -1, dtype=np.int64)
    np.maximum.at(top, rid, packed)
    has = top >= 0
    sv = top >> 34
    best_score = np.where(has & (sv > 0), sv - (1 << 20), best_score).astype(np.int32)
    best_pos = np.where(has, top & ((1 << 33) - 1), best_pos)
    best_rc = np.where(has, (top >> 33) & 1, best_rc).astype(np.uint8)
    if second is not None:
        order = np.lexsort((-packed, rid))                     # per read, candidates in descending key order
        r_s, p_s = rid[order], packed[order]
        bounds = np.searchsorted(r_s, np.arange(R + 1))
        sec_score = np.full(R, SCORE_MIN, dtype=np.int32); sec_pos = np.full(R, -1, dtype=np.int64)
        sec_rc = np.zeros(R, dtype=np.uint8); mapq = np.zeros(R, dtype=np.uint8)
        for r in range(R):
            lo, hi = bounds[r], bounds[r + 1]
            if lo == hi:
                continue
            k = p_s[lo:hi]
            sc_r = ((k >> 34) - (1 << 20)).astype(np.int32)
            out = O.score_reduce(sc_r, (k & ((1 << 33) - 1)).astype(np.uint32), ((k >> 33) & 1).astype(np.uint8), M, second["min_score"] - 1)
            if out[4]:
                sec_score[r], sec_pos[r], sec_rc[r] = out[5], out[6], out[7]
            mapq[r] = O.mapq(second.get("version", 2), second["monotone"], second["perfect_score"], second["min_score"],
                             int(best_score[r]), bool(out[4]), out[5])
        return best_score, best_pos, best_rc, len(keys), dict(second_score=sec_score, second_pos=sec_pos, second_rc=sec_rc, mapq=mapq)
    if want_loci:
        best_g = np.full(R, -1, dtype=np.int64)
        win = packed == top[rid]
        np.maximum.at(best_g, rid[win], g_pos[win].astype(np.int64))
        return best_score, best_pos, best_rc, len(keys), best_g
    if traceback_stride:
        best_wb = np.full(R, -1, dtype=np.int64)
        win = packed == top[rid]
        np.maximum.at(best_wb, rid[win], wb[win].astype(np.int64))
        ids = np.nonzero(best_wb >= 0)[0]
        if traceback_min_score is not None:               # only reads that are aligned (nvBowtie traces valid alignments only)
            ids = np.nonzero((best_wb >= 0) & (best_score >= traceback_min_score))[0]
        twb = best_wb[ids]
        twe = np.minimum(twb + band + M, genome_len)
        sc, src, snk, cig, ln = O.banded_gotoh_traceback_packed_batch(
            band, aln_type, scheme, reads4, roffs, genome2, twb.astype(np.uint32), twe.astype(np.uint32), traceback_stride,
            read_id=ids.astype(np.uint32), flags=(best_rc[ids] * 3).astype(np.uint8), quals=quals)
        tb = dict(ids=ids, scores=sc, pos=twb + src[:, 0].astype(np.int64), sources=src, sinks=snk, cigars=cig, lens=ln)
        return best_score, best_pos, best_rc, len(keys), tb
    return best_score, best_pos, best_rc, len(keys)



def max_text_gaps(scheme, min_score, pattern_len):
    """aln::max_text_gaps, Gotoh aligner (nvbio/alignment/utils_inl.h:141-162)"""
    score = pattern_len * scheme.match
    if score < min_score:
        return 0
    score += scheme.txt_gap_open
    n = 0
    while score >= min_score and n < pattern_len:
        score += scheme.txt_gap_ext
        n += 1
    return (n - 1) & 0xFFFFFFFF


def opposite_window(g, anchor_fw, a_len, o_gapped_len, anchor, genome_len, policy=1, min_frag=0, max_frag=500, overlap=True):
    """BestOppositeScoreStream::init_context (nvBowtie score_inl.h:389-425) + frame_opposite_mate
    (alignment_utils.h:52-88), one hit: (begin, end, opposite_is_fw, valid)"""
    anchor_1 = anchor == 0
    if policy == 0:
        left, fw = (anchor_1 != anchor_fw), anchor_fw
    elif policy == 3:
        left, fw = (anchor_1 == anchor_fw), anchor_fw
    elif policy == 1:
        left, fw = (not anchor_fw), (not anchor_fw)
    else:
        left, fw = anchor_fw, (not anchor_fw)
    if left:
        max_end = g + a_len + o_gapped_len - min_frag if g + a_len + o_gapped_len > min_frag else 0
        begin = g + a_len - max_frag if g + a_len > max_frag else 0
        end = g + a_len if overlap else g
        end = min(end, max_end)
    else:
        min_begin = g + min_frag - o_gapped_len if g + min_frag > o_gapped_len else 0
        end = g + max_frag
        begin = g if overlap else g + a_len
        begin = max(begin, min_begin)
    end = min(end, genome_len)
    return begin, end, fw, (begin < genome_len and begin < end)


def paired_end_cpu(O, hidx, text, genome_len, mates1, mates2, scheme, min_score_of, aln_type, max_frag=500, cigar_stride=0, band=31):
    """the composition of nvbio-gpl_amd/pipeline.py:paired_end on the oracle's functions"""
    R = len(mates1)
    worst = -(1 << 30)
    cand = []
    state = []
    for anchor, (a, o) in enumerate(((mates1, mates2), (mates2, mates1))):
        a_len, o_len = a.shape[1], o.shape[1]
        bs, bp, brc, nc, bg = seed_and_extend_cpu(O, hidx, text, genome_len, a, aln_type=aln_type, scheme=scheme, want_loci=True)
        o_win = np.full((R, 2), -1, dtype=np.int64)
        o_min = min_score_of(o_len)
        gaps = max_text_gaps(scheme, o_min, o_len)
        o_score = np.full(R, worst, dtype=np.int64); o_pos = np.full(R, -1, dtype=np.int64); o_rc = np.zeros(R, dtype=np.uint8)
        for r in range(R):
            if bg[r] < 0 or bs[r] < min_score_of(a_len):
                continue
            begin, end, fw, valid = opposite_window(int(bg[r]), brc[r] == 0, a_len, o_len + gaps, anchor, genome_len, max_frag=max_frag)
            if not valid:
                continue
            p = o[r] if fw else np.where(o[r][::-1] < 4, 3 - o[r][::-1], o[r][::-1]).astype(np.uint8)
            ok, s_, k_ = O.full_gotoh(aln_type, 0, scheme, p, text[begin:end], None, o_min)
            if s_ >= o_min:
                o_score[r] = s_; o_pos[r] = begin + k_[0]; o_rc[r] = 0 if fw else 1
                o_win[r] = (begin, end)
        cand.append((bs.astype(np.int64), bp, brc, o_score, o_pos, o_rc))
        state.append((bg, o_win))
    (s1a, p1a, r1a, s2a, p2a, r2a), (s2b, p2b, r2b, s1b, p1b, r1b) = cand
    pair_a = np.where(s2a > worst, s1a + s2a, worst)
    pair_b = np.where(s1b > worst, s2b + s1b, worst)
    use_b = pair_b > pair_a
    paired = (pair_a > worst) | (pair_b > worst)
    out = dict(anchor=np.where(paired, use_b.astype(np.int64), -1),
               score1=np.where(use_b, s1b, s1a), pos1=np.where(use_b, p1b, p1a), rc1=np.where(use_b, r1b, r1a),
               score2=np.where(use_b, s2b, s2a), pos2=np.where(use_b, p2b, p2a), rc2=np.where(use_b, r2b, r2a),
               pair_score=np.maximum(pair_a, pair_b))
    if not cigar_stride:
        return out
    for m in (1, 2):
        out["begin%d" % m] = np.full(R, -1, dtype=np.int64)
        out["cigars%d" % m] = np.zeros((R, cigar_stride), dtype=np.uint16)
        out["cigar_lens%d" % m] = np.zeros(R, dtype=np.uint32)

    def oriented(read, rc):
        return np.where(read[::-1] < 4, 3 - read[::-1], read[::-1]).astype(np.uint8) if rc else read

    def put(m, r, begin, cig):
        out["begin%d" % m][r] = begin
        k = min(len(cig), cigar_stride)
        out["cigars%d" % m][r, :k] = cig[:k]; out["cigar_lens%d" % m][r] = len(cig)

    half = band // 2
    for r in range(R):
        anchor = int(out["anchor"][r])
        if anchor < 0:
            continue
        (a, o) = (mates1, mates2) if anchor == 0 else (mates2, mates1)
        am, om = (1, 2) if anchor == 0 else (2, 1)
        bg, o_win = state[anchor]
        g = int(bg[r]); wb = g - half if g > half else 0; we = min(wb + band + a.shape[1], genome_len)
        _, _, src, _, cig, _ = O.banded_gotoh_traceback(band, aln_type, scheme, oriented(a[r], out["rc%d" % am][r]), text[wb:we])
        put(am, r, wb + src[0], cig)
        begin, end = int(o_win[r, 0]), int(o_win[r, 1])
        _, _, src, _, cig = O.full_gotoh_traceback(aln_type, scheme, oriented(o[r], out["rc%d" % om][r]), text[begin:end], None,
                                                   min_score_of(o.shape[1]))
        put(om, r, begin + src[0], cig)
    return out



def nvbowtie_best_approx_batch_cpu(O, hidx, text, genome_len, reads, scheme, aln_type, min_score, seed_len=22, seed_freq=None, max_hits=100,
                                   rep_seeds=1000, max_effort=15, max_effort_init=15, min_ext=30, max_ext=400, max_reseed=2, band=31, top_seed=0,
                                   batch_size=None, multi_hit=True):
    """The same loop run as the reference runs it -- pass by pass over the whole batch -- with its several-hits-per-read phase
    (aligner_best_approx.h:487-510: once at most BATCH_SIZE / 2 reads are active every read selects up to n = min( BATCH_SIZE / active,
    min( 4096, max_ext - n_ext ) ) SA rows per pass, select_multi_kernel select_inl.h:268-437; score_reduce_kernel then walks a read's
    hits in selection order with n_ext + i as hit i's extension count, reduce_inl.h:94-134, reduce.h:82-92; n_ext advances by n).
    The number of hits a read selects in a pass depends on how many OTHER reads are still active, so this cannot be run read by read.
    Test infrastructure; parity unpinned beyond the pieces that are pinned on their own."""
    R, M = reads.shape
    L = min(seed_len, M)
    S = seed_freq or int(np.float32(1.0) + np.float32(1.15) * np.sqrt(np.float32(M)))
    retry_stride = S // (max_reseed + 1)
    max_effort_init = max(max_effort_init, max_effort); max_ext = max(max_ext, max_effort)
    BATCH = batch_size or R
    stored = reads[:, ::-1]
    best = np.zeros((R, 6), dtype=np.int64)
    best[:, 0] = min_score; best[:, 3] = min_score; best[:, 1] = 0xFFFFFFFF; best[:, 4] = 0xFFFFFFFF
    n_extensions = passes = multi_passes = 0
    queue = list(range(R))
    for seeding_pass in range(max_reseed + 1):
        if not queue:
            break
        first = seeding_pass * retry_stride
        if M < L + first:
            break
        spr = (M - L - first) // S + 1
        seed_off = first + np.arange(spr) * S
        deques, trys, nxt = {}, {}, []
        for r in queue:
            seeds = np.concatenate([stored[r, o:o + L] for o in seed_off]).astype(np.uint8)
            offs = (np.arange(spr + 1) * L).astype(np.uint32)
            fw = O.match_batch(hidx, seeds, offs, reverse=True)
            comp = np.where(seeds < 4, 3 - seeds, seeds).astype(np.uint8)
            rc = O.match_batch(hidx, comp, offs)
            deques[r], reseed = O.map_exact_read(fw, rc, seed_off, M, L, max_hits, rep_seeds)
            trys[r] = max_effort_init
            if reseed:
                nxt.append(r)
        active = [(r, top_seed) for r in queue]
        n_ext = 0
        while active and n_ext < max_ext:
            n_multi = 1
            if multi_hit and len(active) <= BATCH // 2:
                n_multi = max(1, min(BATCH // len(active), min(4096, max_ext - n_ext)))
            out = []
            for r, top in active:
                if trys[r] == 0 or len(deques[r]) == 0:
                    if trys[r] != 0:
                        deques[r] = deques[r][:0]
                    continue
                hits = []
                for _ in range(n_multi):
                    ok, row, seed, top, deques[r] = O.select_read(deques[r], top)
                    if not ok:
                        break
                    hits.append((row, seed))
                if hits:
                    out.append((r, top, hits))
            if not out:
                break
            for r, top, hits in out:
                b = list(best[r]); erase_any = False
                for idx, (row, seed) in enumerate(hits):
                    pos = int(O.locate_batch(hidx, np.array([row], dtype=np.uint32))[0])
                    g_pos = (pos - (seed & 0xFFF)) & 0xFFFFFFFF
                    read_rc = (seed >> 13) & 1
                    begin = g_pos - band // 2 if g_pos > band // 2 else 0
                    end = min((begin + band + M) & 0xFFFFFFFF, genome_len)
                    pat = (np.where(reads[r, ::-1] < 4, 3 - reads[r, ::-1], reads[r, ::-1]) if read_rc else reads[r]).astype(np.uint8)
                    if end > begin:
                        _, score, _ = O.banded_gotoh(band, aln_type, scheme, pat, text[begin:end])
                    else:
                        score = SCORE_MIN
                    score = max(score, -65536)
                    b, trys[r], erase = O.score_reduce_effort(b, trys[r], score, g_pos, read_rc, (seed >> 14) & 1, M, n_ext + idx, max_effort, min_ext, max_ext)
                    erase_any = erase_any or erase
                best[r] = b
                if erase_any:
                    deques[r] = deques[r][:0]
                n_extensions += len(hits)
            n_ext += n_multi; passes += 1; multi_passes += (n_multi > 1)
            active = [(r, top) for r, top, _ in out]
        queue = nxt
    return dict(best_score=best[:, 0].astype(np.int32), best_loc=np.where(best[:, 1] == 0xFFFFFFFF, -1, best[:, 1]), best_rc=best[:, 2].astype(np.uint8),
                second_score=best[:, 3].astype(np.int32), second_loc=np.where(best[:, 4] == 0xFFFFFFFF, -1, best[:, 4]),
                second_rc=best[:, 5].astype(np.uint8), n_extensions=n_extensions, passes=passes, multi_passes=multi_passes)


def nvbowtie_best_approx_paired_cpu(O, hidx, text, genome_len, mates1, mates2, scheme, aln_type, min_score_of, seed_len=22, seed_freq=None, max_hits=100,
                                    rep_seeds=1000, max_effort=15, max_effort_init=15, min_ext=30, max_ext=400, max_reseed=2, band=31, top_seed=0,
                                    batch_size=None, multi_hit=True, policy=1, min_frag=0, max_frag=500, overlap=True, unpaired=True):
    """nvBowtie's PAIRED-END best-approx loop (Aligner::best_approx, aligner_best_approx_paired.h:84-200,590-1000), pass by pass over the batch:
    for anchor = mate 1, then mate 2, the single-end loop over the anchor's seed hits, a selected hit scored as a pair --
    BestAnchorScoreStream (score_inl.h:143-274): the anchor band-aligned against max( target_pair - perfect( opposite ), min_score( anchor ) ) + 1,
    target_pair = compute_target_score (alignment_utils.h:93-102: the sum of the two min scores until a second pair exists, then
    second + 3/4 (best - second)); BestOppositeScoreStream (score_inl.h:283-456): full-matrix DP of the other mate in its fragment window for the
    hits whose anchor passed; score_reduce_paired_kernel (reduce_inl.h:157-290) with the effort context -- including what the reference's
    code does rather than intends: init_alignments passes `mate` where io::Alignment takes `rc`; the unpaired update writes memory but not the
    kernel's local copy.  Not reproduced: BestAnchorScoreStream::init_context's read of context->min_score before it is assigned (taken as false).
    Alignment = [score, pos, sink, rc, mate, paired].  Test infrastructure; parity unpinned (device-only sources)."""
    R = len(mates1)
    reads = (mates1, mates2)
    lens = (mates1.shape[1], mates2.shape[1])
    worst = (min_score_of(lens[0]), min_score_of(lens[1]))
    BATCH = batch_size or R
    max_effort_init = max(max_effort_init, max_effort); max_ext = max(max_ext, max_effort)
    NONE = 0xFFFFFFFF
    best_a = [[[worst[0], NONE, 255, 0, 0, 0], [worst[0], NONE, 255, 0, 0, 0]] for _ in range(R)]
    best_o = [[[worst[1], NONE, 255, 1, 0, 0], [worst[1], NONE, 255, 1, 0, 0]] for _ in range(R)]

    def paired(a): return a[5] == 1 and a[1] != NONE
    def best_score(b): return b[0][0] + (b[2][0] if paired(b[0]) else 0)
    def second_score(b): return b[1][0] + (b[3][0] if paired(b[1]) else 0)

    def target_score(b, aw, ow):
        if not paired(b[1]):
            return aw + ow
        delta = best_score(b) - second_score(b)
        return second_score(b) + int(delta * 3 / 4) if delta * 3 >= 0 else second_score(b) - ((-delta * 3) // 4)      # C division truncates toward zero

    def visited(b, mate, rc, g):
        return any(mate == x[4] and rc == x[3] and g == x[1] for x in b)

    def frame(anchor, anchor_fw):
        a1 = anchor == 0
        if policy == 0: return (a1 != anchor_fw), anchor_fw
        if policy == 3: return (a1 == anchor_fw), anchor_fw
        if policy == 1: return (not anchor_fw), (not anchor_fw)
        return anchor_fw, (not anchor_fw)

    def oriented(read, rc):
        return (np.where(read[::-1] < 4, 3 - read[::-1], read[::-1]) if rc else read).astype(np.uint8)

    def pair_mate(a, o, m): return a if m == a[4] else o

    def pairs_distinct(a1, o1, a2, o2, dist=None):
        p10, p11, p20, p21 = pair_mate(a1, o1, 0), pair_mate(a1, o1, 1), pair_mate(a2, o2, 0), pair_mate(a2, o2, 1)
        u = lambda v: v & 0xFFFFFFFF
        ap1, op1, ap2, op2 = u(p10[1] + p10[2]), u(p11[1] + p11[2]), u(p20[1] + p20[2]), u(p21[1] + p21[2])
        if dist is None:
            return p10[3] != p20[3] or p11[3] != p21[3] or ap1 != ap2 or op1 != op2
        if p10[3] != p20[3] or p11[3] != p21[3]:
            return True
        near = lambda x, y: x >= y - min(y, dist) and x <= y + dist
        return not (near(ap1, ap2) and near(op1, op2))

    def distinct(pos1, rc1, pos2, rc2, dist):
        return True if rc1 != rc2 else not (pos1 >= pos2 - min(pos2, dist) and pos1 <= pos2 + dist)

    n_extensions = n_opposite = passes = 0
    for anchor in (0, 1):
        a_reads, o_reads = reads[anchor], reads[1 - anchor]
        M, Mo = lens[anchor], lens[1 - anchor]
        a_worst, o_worst = worst[anchor], worst[1 - anchor]
        a_opt, o_opt = scheme.match * M, scheme.match * Mo
        L = min(seed_len, M)
        S = seed_freq or int(np.float32(1.0) + np.float32(1.15) * np.sqrt(np.float32(M)))
        retry_stride = S // (max_reseed + 1)
        stored = a_reads[:, ::-1]
        queue = list(range(R))
        for seeding_pass in range(max_reseed + 1):
            if not queue:
                break
            first = seeding_pass * retry_stride
            if M < L + first:
                break
            spr = (M - L - first) // S + 1
            seed_off = first + np.arange(spr) * S
            deques, trys, nxt = {}, {}, []
            for r in queue:
                seeds = np.concatenate([stored[r, o:o + L] for o in seed_off]).astype(np.uint8)
                offs = (np.arange(spr + 1) * L).astype(np.uint32)
                fw = O.match_batch(hidx, seeds, offs, reverse=True)
                comp = np.where(seeds < 4, 3 - seeds, seeds).astype(np.uint8)
                rc = O.match_batch(hidx, comp, offs)
                deques[r], reseed = O.map_exact_read(fw, rc, seed_off, M, L, max_hits, rep_seeds)
                trys[r] = max_effort_init
                if reseed:
                    nxt.append(r)
            active = [(r, top_seed) for r in queue]
            n_ext = 0
            while active and n_ext < max_ext:
                n_multi = 1
                if multi_hit and len(active) <= BATCH // 2:
                    n_multi = max(1, min(BATCH // len(active), min(4096, max_ext - n_ext)))
                out = []
                for r, top in active:
                    if trys[r] == 0 or len(deques[r]) == 0:
                        continue
                    hits = []
                    for _ in range(n_multi):
                        ok, row, seed, top, deques[r] = O.select_read(deques[r], top)
                        if not ok:
                            break
                        hits.append((row, seed))
                    if hits:
                        out.append((r, top, hits))
                if not out:
                    break
                # score every selected hit against the state BEFORE this pass's reduction (the streams run before score_reduce)
                scored = []
                for r, top, hits in out:
                    bb = [best_a[r][0], best_a[r][1], best_o[r][0], best_o[r][1]]          # [a1, a2, o1, o2]
                    rows = []
                    for row, seed in hits:
                        pos = int(O.locate_batch(hidx, np.array([row], dtype=np.uint32))[0])
                        g = (pos - (seed & 0xFFF)) & 0xFFFFFFFF
                        rc = (seed >> 13) & 1
                        tp = target_score(bb, a_worst, o_worst)
                        tm = max(tp - o_opt, a_worst)
                        skip = visited(bb, anchor, rc, g)
                        begin = g - band // 2 if g > band // 2 else 0
                        end = min((begin + band + M) & 0xFFFFFFFF, genome_len)
                        ms = max(tm + 1, SCORE_MIN)
                        s1, sink1 = SCORE_MIN, 0xFFFFFFFF
                        if not skip and begin < genome_len and end >= begin and end > begin:
                            _, s1, k1 = O.banded_gotoh(band, aln_type, scheme, oriented(a_reads[r], rc), text[begin:end])
                            sink1 = k1[0]
                        passed = (not skip) and s1 >= ms
                        hit_score = s1 if passed else -65536
                        hit_sink = (begin + sink1) & 0xFFFFFFFF if not (skip or begin >= genome_len or end < begin) else 0xFFFFFFFF
                        o_score, o_loc, o_sink = -65536, 0, 0
                        if passed and hit_score != -65536:
                            n_opposite += 1
                            tm2 = max(tp - hit_score, o_worst)
                            ms2 = max(tm2 + 1, SCORE_MIN)
                            run = not (ms2 > o_opt)
                            o_left, o_fw = frame(anchor, rc == 0)
                            gaps = max_text_gaps(scheme, ms2, Mo)
                            gaps = gaps - (1 << 32) if gaps >= (1 << 31) else gaps
                            og = (Mo + gaps) & 0xFFFFFFFF
                            if o_left:
                                max_end = g + M + og - min_frag if g + M + og > min_frag else 0
                                obeg = g + M - max_frag if g + M > max_frag else 0
                                oend = g + M if overlap else g
                                oend = min(oend, max_end)
                            else:
                                min_begin = g + min_frag - og if g + min_frag > og else 0
                                oend = g + max_frag
                                obeg = g if overlap else g + M
                                obeg = max(obeg, min_begin)
                            oend = min(oend, genome_len)
                            if obeg >= genome_len:
                                run = False
                            o_rc = 0 if o_fw else 1
                            if visited(bb, 1 - anchor, o_rc, g) or obeg == oend or oend < obeg:
                                run = False
                            o_loc = obeg if run else 0
                            o_sink = o_loc
                            if run:
                                _, s2, k2 = O.full_gotoh(aln_type, 0, scheme, oriented(o_reads[r], o_rc), text[obeg:oend], None, ms2)
                                if s2 >= ms2:
                                    o_score = s2
                                o_sink = (obeg + (k2[0] if k2[0] != 0xFFFFFFFF else 0)) & 0xFFFFFFFF
                        rows.append((g, hit_sink, hit_score, o_score, o_loc, o_sink, rc, (seed >> 14) & 1))
                    scored.append(rows)
                # score_reduce_paired
                for (r, top, hits), rows in zip(out, scored):
                    a1, a2, o1, o2 = [list(x) for x in (best_a[r][0], best_a[r][1], best_o[r][0], best_o[r][1])]
                    m = [list(a1), list(a2), list(o1), list(o2)]              # what has been written to memory
                    tr = trys[r]; erase = False

                    def failure(idx, top_flag):
                        nonlocal tr
                        if tr > 0:
                            if n_ext + idx >= min_ext and top_flag == 0:
                                tr -= 1
                                if tr == 0:
                                    return True
                            if n_ext + idx >= max_ext:
                                return True
                        return False

                    for idx, (gx, gy, s1, s2, ox, oy, rc, top_flag) in enumerate(rows):
                        score = s1 + s2
                        o_left, o_fw = frame(anchor, rc == 0)
                        o_rc = 0 if o_fw else 1
                        pa = [s1, gx, (gy - gx) & 0xFFFFFFFF, rc, anchor, 1]
                        po = [s2, ox, (oy - ox) & 0xFFFFFFFF, o_rc, 1 - anchor, 1]
                        bl = [a1, a2, o1, o2]
                        if (not pairs_distinct(a1, o1, pa, po)) or (not pairs_distinct(a2, o2, pa, po)):
                            continue
                        if score > best_score(bl):
                            tr = max_effort
                            a2, o2 = a1, o1; a1, o1 = pa, po
                            m = [list(a1), list(a2), list(o1), list(o2)]
                        elif score > second_score(bl) and pairs_distinct(a1, o1, pa, po, M // 2):
                            tr = max_effort
                            a2, o2 = pa, po
                            m = [list(a1), list(a2), list(o1), list(o2)]
                        elif unpaired and not paired(a1):
                            m1 = a1 if anchor == a1[4] else o1
                            m2 = a2 if anchor == a2[4] else o2
                            m1, m2 = list(m1), list(m2)
                            wrote = False
                            if s1 > m1[0]:
                                m2 = m1; m1 = [s1, gx, (gy - gx) & 0xFFFFFFFF, rc, anchor, 0]; wrote = True
                            elif s1 > m2[0] and distinct(m1[1], m1[3], gx, rc, M // 2):
                                m2 = [s1, gx, (gy - gx) & 0xFFFFFFFF, rc, anchor, 0]; wrote = True
                            elif failure(idx, top_flag):
                                erase = True
                            if wrote:
                                if anchor:
                                    m[2], m[3] = list(m1), list(m2)
                                else:
                                    m[0], m[1] = list(m1), list(m2)
                        elif failure(idx, top_flag):
                            erase = True
                    best_a[r] = [m[0], m[1]]; best_o[r] = [m[2], m[3]]
                    trys[r] = tr
                    if erase:
                        deques[r] = deques[r][:0]
                    n_extensions += len(rows)
                n_ext += n_multi; passes += 1
                active = [(r, top) for r, top, _ in out]
            queue = nxt
    return dict(best_a=np.array(best_a, dtype=np.int64), best_o=np.array(best_o, dtype=np.int64), n_extensions=n_extensions, n_opposite=n_opposite, passes=passes)

# ---- nvBowtie's scoring stream, restated (test infrastructure; parity unpinned: score_inl.h is device-only CUDA) ------------------
def score_stream_flatten(idx_queue, hit_read_id, hit_seed, hit_loc, read_index, band_len, genome_len, reads_reversed=True):
    """BestScoreStream::init_context (nvBowtie/bowtie2/cuda/score_inl.h:85-115) and the read orientation load_strings requests
    (alignment_utils.h:291-296: `read_rc ? FORWARD : REVERSE`, `read_rc ? COMPLEMENT : STANDARD` over reads stored reversed),
    one work item at a time.  packed_seed (defs.h:162-172): pos_in_read:12, index_dir:1, rc:1, top_flag:1 from bit 0.
    Returns (read_id, flags, genome_begin, genome_end) as the per-job arrays of an alignment batch
    (flags: 1 = read the stored stream reversed, 2 = complemented)."""
    n = len(idx_queue) if idx_queue is not None else len(hit_read_id)
    rid = np.zeros(n, dtype=np.uint32); flags = np.zeros(n, dtype=np.uint8)
    wb = np.zeros(n, dtype=np.uint32); we = np.zeros(n, dtype=np.uint32)
    for i in range(n):
        idx = int(idx_queue[i]) if idx_queue is not None else i                      # context->idx = idx_queue[i]             (:89)
        read_id = int(hit_read_id[idx])                                               # hit.read_id                            (:97)
        read_rc = (int(hit_seed[idx]) >> 13) & 1                                      # hit.seed.rc                            (:96)
        g_pos = int(hit_loc[idx])                                                     # hit.loc                                (:100)
        read_len = int(read_index[read_id + 1]) - int(read_index[read_id])            # read_range.y - read_range.x            (:102)
        begin = g_pos - band_len // 2 if g_pos > band_len // 2 else 0                 # genome_begin                           (:103)
        end = min((begin + band_len + read_len) & 0xFFFFFFFF, genome_len)             # genome_end (uint32 arithmetic)         (:104)
        if begin >= genome_len or end < begin:     # a wrapped locus (seed hanging over the genome start) makes the reference read out of bounds:
            begin = end = 0                        # here the job gets the empty window [0, 0) and reports nothing
        if reads_reversed:
            f = 2 if read_rc else 1                                                   # FORWARD + COMPLEMENT : REVERSE + STANDARD
        else:
            f = 3 if read_rc else 0
        rid[i], flags[i], wb[i], we[i] = read_id, f, begin, end
    return rid, flags, wb, we


def score_stream_output(idx_queue, n_hits, scores, sinks, genome_begin, worst_score=-65536):
    """BestScoreStream::output (score_inl.h:119-133): hit.score = max(sink.score, worst_score); hit.sink = genome_begin + sink.sink.x"""
    hit_score = np.zeros(n_hits, dtype=np.int32); hit_sink = np.zeros(n_hits, dtype=np.uint32)
    for i in range(len(scores)):
        idx = int(idx_queue[i]) if idx_queue is not None else i
        hit_score[idx] = max(int(scores[i]), worst_score)
        hit_sink[idx] = (int(genome_begin[i]) + int(sinks[i][0])) & 0xFFFFFFFF
    return hit_score, hit_sink


def nvbowtie_best_approx_cpu(O, hidx, text, genome_len, reads, scheme, aln_type, min_score, seed_len=22, seed_freq=None, max_hits=100,
                             rep_seeds=1000, max_effort=15, max_effort_init=15, min_ext=30, max_ext=400, max_reseed=2, band=31, top_seed=0):
    """nvBowtie's best-approx single-end loop, one read at a time, on the oracle (Aligner::best_approx, aligner_best_approx.h:39-207,
    363-667; map_kernel mapping_inl.h:485-556; select_kernel select_inl.h:62-130; locate locate_inl.h:113-138; BestScoreStream
    score_inl.h:85-133; score_reduce_kernel reduce_inl.h:65-140).  reads: uint8 [R, M] in their ORIGINAL orientation; nvBowtie stores
    them reversed and seeds the stored stream.  One hit per read and pass.  Test infrastructure; parity unpinned beyond the pieces
    that are pinned on their own (match / locate / banded DP / the deque container)."""
    R, M = reads.shape
    L = min(seed_len, M)
    S = seed_freq or int(1 + 1.15 * math.sqrt(M))
    retry_stride = S // (max_reseed + 1)
    max_effort_init = max(max_effort_init, max_effort); max_ext = max(max_ext, max_effort)
    stored = reads[:, ::-1]
    best = np.zeros((R, 6), dtype=np.int64)
    best[:, 0] = min_score; best[:, 3] = min_score; best[:, 1] = 0xFFFFFFFF; best[:, 4] = 0xFFFFFFFF
    n_extensions = 0
    queue = list(range(R))
    for seeding_pass in range(max_reseed + 1):
        if not queue:
            break
        first = seeding_pass * retry_stride
        if M < L + first:
            break
        spr = (M - L - first) // S + 1
        seed_off = first + np.arange(spr) * S
        nxt = []
        for r in queue:
            seeds = np.concatenate([stored[r, o:o + L] for o in seed_off]).astype(np.uint8)
            offs = (np.arange(spr + 1) * L).astype(np.uint32)
            fw = O.match_batch(hidx, seeds, offs, reverse=True)                         # forward scan of the stored seed
            comp = np.where(seeds < 4, 3 - seeds, seeds).astype(np.uint8)
            rc = O.match_batch(hidx, comp, offs)                                       # reverse scan, complemented
            deque, reseed = O.map_exact_read(fw, rc, seed_off, M, L, max_hits, rep_seeds)
            if reseed:
                nxt.append(r)
            trys, top, n_ext = max_effort_init, top_seed, 0
            while n_ext < max_ext:
                if trys == 0:
                    break
                ok, row, seed, top, deque = O.select_read(deque, top)
                if not ok:
                    break
                pos = int(O.locate_batch(hidx, np.array([row], dtype=np.uint32))[0])
                g_pos = (pos - (seed & 0xFFF)) & 0xFFFFFFFF                           # hit.loc (locate_inl.h:133)
                read_rc = (seed >> 13) & 1
                begin = g_pos - band // 2 if g_pos > band // 2 else 0
                end = min((begin + band + M) & 0xFFFFFFFF, genome_len)
                pat = (np.where(reads[r, ::-1] < 4, 3 - reads[r, ::-1], reads[r, ::-1]) if read_rc else reads[r]).astype(np.uint8)
                if end > begin:
                    _, score, _ = O.banded_gotoh(band, aln_type, scheme, pat, text[begin:end])
                else:
                    score = SCORE_MIN
                score = max(score, -65536)                                            # hit.score = max( sink.score, worst_score )
                b, trys, erase = O.score_reduce_effort(list(best[r]), trys, score, g_pos, read_rc, (seed >> 14) & 1, M, n_ext, max_effort, min_ext, max_ext)
                best[r] = b
                if erase:
                    deque = deque[:0]
                n_ext += 1; n_extensions += 1
        queue = nxt
    out = dict(best_score=best[:, 0].astype(np.int32), best_loc=np.where(best[:, 1] == 0xFFFFFFFF, -1, best[:, 1]), best_rc=best[:, 2].astype(np.uint8),
               second_score=best[:, 3].astype(np.int32), second_loc=np.where(best[:, 4] == 0xFFFFFFFF, -1, best[:, 4]),
               second_rc=best[:, 5].astype(np.uint8), n_extensions=n_extensions)
    return out
