"""microbenchmark: random 8-byte gathers from tables of growing size (TLB reach / DRAM request rate)"""
import time, torch
dev = "cuda:0"
Q = 100_000_000
g = torch.Generator(device=dev); g.manual_seed(1)
for gb in (0.125, 0.5, 2, 8, 16, 34, 64):
    n = int(gb * (1 << 30) / 8)
    t = torch.empty(n, dtype=torch.int64, device=dev)
    t.fill_(3)
    idx = torch.randint(0, n, (Q,), device=dev, generator=g)
    out = torch.empty(Q, dtype=torch.int64, device=dev)
    for rep in range(3):
        torch.cuda.synchronize(); t0 = time.time()
        torch.index_select(t, 0, idx, out=out)
        torch.cuda.synchronize(); dt = time.time() - t0
    print("table %6.3f GB: %.2f ms -> %.1f G gathers/s" % (gb, dt * 1e3, Q / dt / 1e9), flush=True)
    del t, idx, out
