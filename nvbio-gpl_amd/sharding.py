"""Read-shard parallelism across the GPUs of one node.

The path has no exchange step: every read / seed / candidate is independent, the index is
replicated per GPU (about 3 GB for a 3 Gbp reference, against 288 GB of HBM each), and a batch
of reads is split into contiguous per-rank ranges.  The only communication is the final gather of
the per-read result -- (best score, end position, strand) packed into 8 bytes per read -- to rank 0,
one RCCL gather over xGMI per batch (backend "nccl" is RCCL on ROCm; "gloo" on CPU in tests),
issued asynchronously so that it overlaps the next batch's kernels (ResultGatherer).
The reference itself is single-GPU (nvBowtie/nvBowtie.cpp:224-263): this is new design, not a
translation of a call site.
"""


def shard_bounds(total, world, rank):
    """contiguous range [begin, end) of `total` reads owned by `rank` (balanced to within one read)"""
    base, rem = divmod(int(total), int(world))
    begin = rank * base + min(rank, rem)
    return begin, begin + base + (1 if rank < rem else 0)


def pack_result(best_score, best_pos, best_rc):
    """[R, 2] int64: (score, pos * 2 + rc); pos == -1 (unaligned) stays negative"""
    import torch
    return torch.stack([best_score.to(torch.int64), best_pos.to(torch.int64) * 2 + best_rc.to(torch.int64)], dim=1)


def unpack_result(packed):
    score = packed[:, 0].to(packed.dtype)
    pr = packed[:, 1]
    return score, pr >> 1, pr & 1


SCORE_BIAS = 1 << 20
SCORE_MIN = -(1 << 30)


def pack_result64(best_score, best_pos, best_rc):
    """[R] int64, 8 bytes per read: (score + 2^20) << 35 | (pos + 1) << 1 | rc.  pos is a uint32 text
    position or -1 (unaligned); scores of aligned reads lie within +-2^20 (io::Alignment keeps 18 bits)."""
    import torch
    s = torch.clamp(best_score.to(torch.int64) + SCORE_BIAS, min=0)
    return (s << 35) | ((best_pos.to(torch.int64) + 1) << 1) | best_rc.to(torch.int64)


def unpack_result64(packed):
    import torch
    pos = ((packed >> 1) & ((1 << 34) - 1)) - 1
    score = (packed >> 35) - SCORE_BIAS
    score = torch.where(pos >= 0, score, torch.full_like(score, SCORE_MIN))
    return score, pos, packed & 1


def gather_results(dist, packed, world, rank, dst=0, sizes=None):
    """gather ragged per-rank results (any trailing shape) to `dst`; returns the concatenation in read order
    on dst, None elsewhere.  sizes: per-rank row counts if the caller knows them (no size exchange then)."""
    import torch
    if world == 1:
        return packed
    if sizes is None:
        szt = [torch.zeros(1, dtype=torch.int64, device=packed.device) for _ in range(world)]
        dist.all_gather(szt, torch.tensor([packed.shape[0]], dtype=torch.int64, device=packed.device))
        sizes = [int(s.item()) for s in szt]
    mx = max(sizes)
    if packed.shape[0] == mx:
        buf = packed.contiguous()
    else:
        buf = torch.zeros((mx,) + tuple(packed.shape[1:]), dtype=packed.dtype, device=packed.device)
        buf[:packed.shape[0]] = packed
    out = [torch.empty_like(buf) for _ in range(world)] if rank == dst else None
    dist.gather(buf, out, dst=dst)
    if rank != dst:
        return None
    return torch.cat([o[:s] for o, s in zip(out, sizes)], dim=0)


class ResultGatherer:
    """The per-batch result gather, overlapped with the next batch: submit() enqueues an asynchronous gather
    of this rank's packed results (equal shard sizes, known to every rank) into one of two receive buffers on
    `dst` and returns at once; the collective runs on the backend's own stream after the kernels that
    produced `packed`, while the caller launches the next batch.  wait() blocks until everything submitted
    has landed; result(k) is the [world, R] receive buffer of the k-th most recent batch on dst (read order =
    rank-major), valid until two more batches are submitted."""

    def __init__(self, dist, world, rank, rows, device, dst=0, dtype=None):
        import torch
        self.dist, self.world, self.rank, self.dst = dist, world, rank, dst
        dtype = dtype or torch.int64
        self.recv = [[torch.empty(rows, dtype=dtype, device=device) for _ in range(world)] for _ in range(2)] if rank == dst else None
        self.pending = []          # (work handle, the send buffer it reads: kept alive until the work is done)
        self.k = 0

    def submit(self, packed):
        if len(self.pending) >= 2:                       # the buffer about to be reused must have been filled
            self.pending.pop(0)[0].wait()
        out = self.recv[self.k & 1] if self.rank == self.dst else None
        work = self.dist.gather(packed, out, dst=self.dst, async_op=True)
        self.pending.append((work, packed))
        self.k += 1

    def wait(self):
        for w, _ in self.pending:
            w.wait()
        self.pending = []

    def result(self, back=0):
        return self.recv[(self.k - 1 - back) & 1] if self.rank == self.dst else None
