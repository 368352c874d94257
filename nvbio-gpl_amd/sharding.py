"""Read-shard parallelism across the GPUs of one node.

The path has no exchange step: every read / seed / candidate is independent, the index is
replicated per GPU (about 3 GB for a 3 Gbp reference, against 288 GB of HBM each), and a batch
of reads is split into contiguous per-rank ranges.  The only communication is the final gather of
the per-read result -- (best score, end position * 2 + strand), 16 bytes per read -- to rank 0,
one RCCL gather over xGMI per batch (backend "nccl" is RCCL on ROCm; "gloo" on CPU in tests).
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


def gather_results(dist, packed, world, rank, dst=0):
    """gather ragged per-rank results to `dst`; returns the concatenation (read order) on dst, None elsewhere"""
    import torch
    if world == 1:
        return packed
    sizes = [torch.zeros(1, dtype=torch.int64, device=packed.device) for _ in range(world)]
    dist.all_gather(sizes, torch.tensor([packed.shape[0]], dtype=torch.int64, device=packed.device))
    sizes = [int(s.item()) for s in sizes]
    mx = max(sizes)
    buf = torch.zeros((mx, 2), dtype=torch.int64, device=packed.device)
    buf[:packed.shape[0]] = packed
    out = [torch.empty_like(buf) for _ in range(world)] if rank == dst else None
    dist.gather(buf, out, dst=dst)
    if rank != dst:
        return None
    return torch.cat([o[:s] for o, s in zip(out, sizes)], dim=0)
