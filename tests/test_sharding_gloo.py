"""CPU, world_size 2, gloo: the N>1 path (contiguous read shards, no data-path collective, one
gather of per-read results to rank 0) gives exactly the single-process result."""
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import os, sys, importlib
import numpy as np, torch, torch.distributed as dist
sys.path.insert(0, %(root)r)
import __graft_entry__ as ge
import oracle
from oracle import cpu_pipeline
ge.load_package()
sharding = importlib.import_module("nvbio_gpl_amd.sharding")
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
d = np.load(%(data)r)
O = oracle.Oracle()
hidx = oracle.HostIndex(int(d["n"]), int(d["primary"]), d["L2"], d["bwt_occ"], d["ssa"])
b, e = sharding.shard_bounds(len(d["reads"]), world, rank)
bs, bp, brc, nc = cpu_pipeline.seed_and_extend_cpu(O, hidx, d["text"], int(d["n"]), d["reads"][b:e])
packed = sharding.pack_result(torch.from_numpy(bs), torch.from_numpy(bp), torch.from_numpy(brc))
allr = sharding.gather_results(dist, packed, world, rank, dst=0)
# the 8-byte packing and the asynchronous double-buffered gatherer bench.py uses: equal shards, three batches
p64 = sharding.pack_result64(torch.from_numpy(bs), torch.from_numpy(bp), torch.from_numpy(brc))
rows = 700
g = sharding.ResultGatherer(dist, world, rank, rows, "cpu")
for k in range(3):
    g.submit(p64[:rows] + 0 * k if k != 1 else torch.flip(p64[:rows], dims=[0]))
g.wait()
if rank == 0:
    s, p, r = sharding.unpack_result(allr)
    last = torch.cat(g.result(0)); prev = torch.cat(g.result(1))
    s64, p64u, r64 = sharding.unpack_result64(last)
    np.savez(%(out)r, s=s.numpy(), p=p.numpy(), r=r.numpy(), s64=s64.numpy(), p64=p64u.numpy(), r64=r64.numpy(),
             prev=prev.numpy(), last=last.numpy())
else:
    assert allr is None and g.result() is None
dist.barrier(); dist.destroy_process_group()
'''


def test_shard_bounds_cover_everything():
    sys.path.insert(0, ROOT)
    import importlib
    import __graft_entry__ as ge
    ge.load_package()
    sharding = importlib.import_module("nvbio_gpl_amd.sharding")
    for total in (0, 1, 7, 8, 9, 10_000_001):
        for world in (1, 2, 3, 8):
            prev = 0
            for r in range(world):
                b, e = sharding.shard_bounds(total, world, r)
                assert b == prev and e >= b
                prev = e
            assert prev == total


def test_two_rank_gloo_equals_single_process(orc):
    from oracle import cpu_pipeline
    rng = np.random.default_rng(33)
    G = 300000
    text = rng.integers(0, 4, G, dtype=np.uint8)
    hidx = orc.build_index(text)
    R, M = 1501, 150                                    # odd count: ragged shards
    starts = rng.integers(0, G - M, R)
    reads = np.stack([text[s:s + M] for s in starts]).copy()
    reads[rng.random(reads.shape) < 0.02] = 1
    rc = rng.random(R) < 0.5
    reads[rc] = 3 - reads[rc][:, ::-1]
    want = cpu_pipeline.seed_and_extend_cpu(orc, hidx, text, G, reads)
    with tempfile.TemporaryDirectory() as tmp:
        data, out, script = os.path.join(tmp, "d.npz"), os.path.join(tmp, "o.npz"), os.path.join(tmp, "w.py")
        np.savez(data, n=G, primary=hidx.primary, L2=hidx.L2, bwt_occ=hidx.bwt_occ, ssa=hidx.ssa, text=text, reads=reads)
        open(script, "w").write(WORKER % {"root": ROOT, "data": data, "out": out})
        env = dict(os.environ, OMP_NUM_THREADS="2")
        subprocess.check_call([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                               "--master-addr", "127.0.0.1", "--master-port", "29533", script], env=env, timeout=600)
        got = np.load(out)
        assert np.array_equal(got["s"], want[0].astype(np.int64))
        assert np.array_equal(got["p"], want[1])
        assert np.array_equal(got["r"], want[2].astype(np.int64))
        # asynchronous gatherer: batch 3 = the first 700 reads of each shard, batch 2 = the same reversed
        b1 = (R + 1) // 2
        idx = np.concatenate([np.arange(700), b1 + np.arange(700)])
        assert np.array_equal(got["s64"], want[0][idx].astype(np.int64))
        assert np.array_equal(got["p64"], want[1][idx]) and np.array_equal(got["r64"], want[2][idx].astype(np.int64))
        assert np.array_equal(got["prev"].reshape(2, 700)[:, ::-1].reshape(-1), got["last"])
