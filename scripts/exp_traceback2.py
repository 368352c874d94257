import os, sys, json
sys.path.insert(0, "/root/repo")
import torch, bench
import __graft_entry__ as ge
amd = ge.load_package()
dev = "cuda:0"
P, M, n = 10_000_000, 150, 400_000_000
genome = bench.make_reference(n, dev, seed=3)
reads_sym, pos, rc = bench.make_reads(genome, n, P, M, dev, seed=4)
reads_sym = torch.where(rc[:, None], 3 - reads_sym.flip(1), reads_sym)
r4 = bench.pack4(reads_sym.reshape(-1))
roffs = (torch.arange(P + 1, device=dev) * M).to(torch.int32)
wb = torch.clamp(pos - 15, min=0); we = torch.clamp(wb + 31 + M, max=n)
half = torch.arange(P, device=dev) % 2 == 0
al = amd.make_gotoh_aligner(amd.SEMI_GLOBAL, amd.GotohScheme(0, 6, 6, -8, -3, -8, -3))
def t(fn, reps=3):
    out=[]
    for _ in range(reps):
        a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); e.record(); torch.cuda.synchronize(); out.append(a.elapsed_time(e))
    return out
for name, mask in (("all", None), ("half-empty", half)):
    w0 = wb if mask is None else torch.where(mask, wb, torch.zeros_like(wb))
    w1 = we if mask is None else torch.where(mask, we, torch.zeros_like(we))
    b = amd.AlignmentBatch(r4, 4, roffs, genome, 2, w0.to(torch.int32), w1.to(torch.int32), max_read_len=M)
    print(name, "banded tb", t(lambda: amd.BatchedBandedAlignmentTraceback(31, al).enact(b, cigar_stride=16)), flush=True)
    # full: windows 400 wide
    f0 = torch.clamp(pos - 100, min=0); f1 = torch.clamp(f0 + 400, max=n)
    if mask is not None:
        f0 = torch.where(mask, f0, torch.zeros_like(f0)); f1 = torch.where(mask, f1, torch.zeros_like(f1))
    bf = amd.AlignmentBatch(r4, 4, roffs, genome, 2, f0.to(torch.int32), f1.to(torch.int32), max_read_len=M)
    print(name, "full tb", t(lambda: amd.BatchedAlignmentTraceback(al).enact(bf, M, 400, cigar_stride=16)), flush=True)
