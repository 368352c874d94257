"""GPU, world_size 2 on ONE MI355X: two processes, both on cuda:0, each builds its replica of the index and runs the HIP path on
its contiguous shard of the reads; the per-read results travel to rank 0 through the same gather code bench.py uses (gloo over
host copies here: RCCL refuses two ranks on one device) and must equal the single-process run.  What this rehearses before the
first 8-GPU run: N processes of the HIP library side by side (handles, scratch pools, streams), sharding, packing, the gather."""
import importlib
import os
import subprocess
import sys

import numpy as np
import pytest

from util import mutate_reads

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import os, sys, importlib
import numpy as np, torch, torch.distributed as dist
sys.path.insert(0, %(root)r)
import __graft_entry__ as ge
amd = ge.load_package()
pipeline = importlib.import_module("nvbio_gpl_amd.pipeline")
sharding = importlib.import_module("nvbio_gpl_amd.sharding")
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
torch.cuda.set_device(0)                                   # both ranks share the one GPU of the box
d = np.load(%(data)r)
G, M = int(d["G"]), int(d["M"])
genome = torch.from_numpy(d["genome2"].view(np.int32)).cuda()
fmi = amd.FMIndex.build(genome, G, kmer_len=10, sa_int=1)  # every rank holds its own replica of the index
b, e = sharding.shard_bounds(len(d["reads"]), world, rank)
rb = pipeline.ReadBatch(torch.from_numpy(np.load(%(packed)r)["r%%d" %% rank].view(np.int32)).cuda(), e - b, M)     # this rank's shard, 4-bit packed
params = pipeline.SeedExtendParams.end_to_end()
for _ in range(3):                                          # several batches, as the bench's loop
    bs, bp, brc, nc = pipeline.seed_and_extend(fmi, genome, G, rb, params)
torch.cuda.synchronize()
packed = sharding.pack_result64(bs, bp, brc).cpu()          # host copy: gloo gathers host tensors
allr = sharding.gather_results(dist, packed, world, rank, dst=0)
if rank == 0:
    s, p, r = sharding.unpack_result64(allr)
    np.savez(%(out)r, s=s.numpy(), p=p.numpy(), r=r.numpy())
else:
    assert allr is None
fmi.close()
dist.barrier(); dist.destroy_process_group()
'''


@pytest.mark.gpu
def test_two_hip_ranks_on_one_gpu_equal_single_process(amd, orc, tmp_path):
    import torch
    pipeline = importlib.import_module("nvbio_gpl_amd.pipeline")
    sharding = importlib.import_module("nvbio_gpl_amd.sharding")
    rng = np.random.default_rng(71)
    G, R, M = 1_500_000, 40001, 150                          # odd count: ragged shards
    text = rng.integers(0, 4, G, dtype=np.uint8)
    starts = rng.integers(0, G - M - 8, R)
    reads = mutate_reads(rng, text, starts, M)
    rcm = rng.random(R) < 0.5
    reads[rcm] = 3 - reads[rcm][:, ::-1]
    genome2 = orc.pack2(text)
    # single process
    fmi = amd.FMIndex.build(genome2, G, kmer_len=10, sa_int=1)
    g_dev = torch.from_numpy(genome2.view(np.int32)).cuda()
    rb = pipeline.ReadBatch(torch.from_numpy(orc.pack4(reads.reshape(-1)).view(np.int32)).cuda(), R, M)
    bs, bp, brc, _ = pipeline.seed_and_extend(fmi, g_dev, G, rb, pipeline.SeedExtendParams.end_to_end())
    want = (bs.cpu().numpy(), bp.cpu().numpy(), brc.cpu().numpy())
    fmi.close(); del g_dev, rb
    torch.cuda.empty_cache()
    # two ranks
    data, packed, out, script = (str(tmp_path / n) for n in ("d.npz", "p.npz", "o.npz", "w.py"))
    np.savez(data, G=G, M=M, genome2=genome2, reads=reads)
    shards = {}
    for rank in range(2):
        b, e = sharding.shard_bounds(R, 2, rank)
        shards["r%d" % rank] = orc.pack4(reads[b:e].reshape(-1))
    np.savez(packed, **shards)
    open(script, "w").write(WORKER % {"root": ROOT, "data": data, "packed": packed, "out": out})
    env = dict(os.environ, OMP_NUM_THREADS="2", HSA_ENABLE_IPC_MODE_LEGACY="0")
    subprocess.check_call([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                           "--master-addr", "127.0.0.1", "--master-port", "29541", script], env=env, timeout=600)
    got = np.load(out)
    aligned = want[1] >= 0
    assert np.array_equal(got["p"], want[1]) and np.array_equal(got["r"][aligned], want[2][aligned].astype(np.int64))
    assert np.array_equal(got["s"][aligned], want[0][aligned].astype(np.int64))
