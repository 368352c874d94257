import os, sys, importlib, json
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
import __graft_entry__ as ge
amd = ge.load_package()
import bench
dev="cuda:0"
J, M, W, n = 400000, 150, 500, 50_000_000
genome = bench.make_reference(n, dev, seed=5)
g = torch.Generator(device=dev); g.manual_seed(3)
wb = torch.randint(0, n - W - 8, (J,), device=dev, generator=g, dtype=torch.int64)
off = torch.randint(0, W - M - 8, (J,), device=dev, generator=g, dtype=torch.int64)
j = torch.arange(M, device=dev, dtype=torch.int64)[None, :]
has = torch.rand(J, device=dev, generator=g) < 0.5
ip = torch.randint(5, M - 5, (J,), device=dev, generator=g)[:, None]
ig = torch.randint(1, 4, (J,), device=dev, generator=g)[:, None]
src = torch.where(has[:, None] & (j >= ip), j + ig, j)
sym = bench.genome_symbols(genome, (wb + off)[:, None] + src)
sub = torch.rand(J, M, device=dev, generator=g) < 0.01
sym = torch.where(sub, (sym + 1) % 4, sym)
reads4 = bench.pack4(sym.reshape(-1))
roff = torch.arange(J + 1, device=dev, dtype=torch.int32) * M
sv = amd.GotohScheme(0, 6, 6, -8, -3, -8, -3)
al = amd.make_gotoh_aligner(amd.SEMI_GLOBAL, sv)
ids = torch.arange(J, device=dev, dtype=torch.int32)
ms = torch.full((J,), -91, dtype=torch.int32, device=dev)
for variant in ("plain", "read_id", "min_scores", "both"):
    kw = {}
    if variant in ("read_id", "both"): kw["read_id"] = ids
    batch = amd.AlignmentBatch(reads4, 4, roff, genome, 2, wb.to(torch.int32), (wb + W).to(torch.int32), max_read_len=M, **kw)
    msk = ms if variant in ("min_scores", "both") else None
    sc, sk = amd.BatchedAlignmentScore(al, text_blocking=False).enact(batch, M, W, min_scores=msk)
    op = amd.BatchedAlignmentTraceback(al)
    t = []
    for it in range(3):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); out = op.enact(batch, M, W, min_scores=msk, cigar_stride=16, scores=sc, sinks=sk); b.record(); torch.cuda.synchronize()
        t.append(a.elapsed_time(b))
    print(variant, round(sorted(t)[1], 2), "ms", flush=True)
