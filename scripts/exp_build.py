"""experiment: time the GPU index build at benchmark scale and check size-independent properties"""
import sys, time
import numpy as np
import torch
sys.path.insert(0, ".")
import __graft_entry__ as ge
amd = ge.load_package()
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 3_000_000_000
k = int(sys.argv[2]) if len(sys.argv) > 2 else 12
sa_int = int(sys.argv[3]) if len(sys.argv) > 3 else 16
dev = "cuda:0"
g = torch.Generator(device=dev); g.manual_seed(1234)
nw = (n + 15) // 16
t0 = time.time()
words = torch.randint(-2**31, 2**31 - 1, (nw + 4,), dtype=torch.int64, device=dev, generator=g).to(torch.int32)
torch.cuda.synchronize(); print("text gen %.2fs" % (time.time() - t0), flush=True)
t0 = time.time()
fmi = amd.FMIndex.build(words, n, kmer_len=k, sa_int=sa_int)
torch.cuda.synchronize(); print("build n=%d k=%d: %.2fs  owned %.2f GB  peak torch %.1f GB" % (n, k, time.time() - t0, fmi.device_bytes() / 1e9, torch.cuda.max_memory_allocated() / 1e9), flush=True)
free, total = torch.cuda.mem_get_info(); print("free %.1f GB of %.1f" % (free / 1e9, total / 1e9))
v = fmi.view(); print("primary", v.primary, "L2", [v.L2[i] for i in range(5)])
# property: queries cut from the text must hit, and locate must return an occurrence
Q, L = 20_000_000, 22
starts = torch.randint(0, n - L, (Q,), device=dev, generator=g)
idx = (starts[:, None] + torch.arange(L, device=dev)[None, :])
w = words[(idx >> 4)].to(torch.int64) & 0xFFFFFFFF
sym = ((w >> (30 - 2 * (idx & 15))) & 3).to(torch.uint8).contiguous()
qs = amd.PackedStringSet(sym.view(-1), 8, Q, fixed_len=L)
for rep in range(3):
    torch.cuda.synchronize(); t0 = time.time()
    r = fmi.match(qs)
    torch.cuda.synchronize(); dt = time.time() - t0
    print("match 20M x 22 (k=%d): %.3f ms -> %.1f M q/s" % (k, dt * 1e3, Q / dt / 1e6))
for rep in range(2):
    torch.cuda.synchronize(); t0 = time.time()
    r0 = fmi.match(qs, amd.FM_NO_KMER_TABLE)
    torch.cuda.synchronize(); dt = time.time() - t0
    print("match 20M x 22 (no table): %.3f ms -> %.1f M q/s" % (dt * 1e3, Q / dt / 1e6))
assert torch.equal(r, r0)
ru = r.to(torch.int64) & 0xFFFFFFFF
assert bool((ru[:, 0] <= ru[:, 1]).all()), "a substring of the text did not match"
print("range widths: max %d mean %.3f" % (int((ru[:, 1] - ru[:, 0]).max()), float((ru[:, 1] - ru[:, 0] + 1).float().mean())))
torch.cuda.synchronize(); t0 = time.time()
pos = fmi.locate(r[:, 0].contiguous())
torch.cuda.synchronize(); dt = time.time() - t0
print("locate 20M (sa_int=%d): %.3f ms -> %.1f M/s" % (sa_int, dt * 1e3, Q / dt / 1e6))
pu = pos.to(torch.int64) & 0xFFFFFFFF
idx2 = (pu[:, None] + torch.arange(L, device=dev)[None, :])
w2 = words[(idx2 >> 4)].to(torch.int64) & 0xFFFFFFFF
sym2 = ((w2 >> (30 - 2 * (idx2 & 15))) & 3).to(torch.uint8)
assert torch.equal(sym2, sym), "locate() returned a position that does not spell the query"
print("exact hits at the sampled position: %.4f" % float((pu == starts).float().mean()))
print("OK")
