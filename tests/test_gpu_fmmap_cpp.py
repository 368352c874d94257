"""GPU: the C++ fmmap-shaped host program (nvbio-gpl_amd/host/fmmap_amd.cpp -- seeds -> seed pass -> windows -> banded Gotoh ->
best per read over the C ABI, no Python, no torch) returns, read for read, what pipeline.seed_and_extend returns on the same
inputs (and so what the oracle's CPU path returns: tests/test_gpu_pipeline.py)."""
import importlib
import json
import os
import subprocess

import numpy as np
import pytest

from util import mutate_reads

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "nvbio-gpl_amd", "lib", "fmmap_amd")


def test_fmmap_cpp_is_built():
    assert os.path.exists(EXE), "nvbio-gpl_amd/lib/fmmap_amd is missing: __graft_entry__.build() makes it"


@pytest.mark.gpu
@pytest.mark.parametrize("kmer", [10, 15])            # even k: direct table, one seed pass per strand; odd k: canonical table, one pass for both
def test_fmmap_cpp_equals_pipeline(amd, orc, tmp_path, kmer):
    import torch
    pipeline = importlib.import_module("nvbio_gpl_amd.pipeline")
    rng = np.random.default_rng(4)
    G, R, M = 2_000_000, 30000, 150
    text = rng.integers(0, 4, G, dtype=np.uint8)
    text[100000:101000] = text[500000:501000]                      # a repeat: residual seeds
    for c in range(20):
        text[700000 + 3000 * c:700300 + 3000 * c] = text[900000:900300]
    starts = rng.integers(0, G - M - 8, R)
    starts[:200] = rng.integers(100000, 100800, 200)
    starts[200:500] = 700000 + 3000 * rng.integers(0, 20, 300) + rng.integers(0, 150, 300)
    reads = mutate_reads(rng, text, starts, M)
    rcm = rng.random(R) < 0.5
    reads[rcm] = 3 - reads[rcm][:, ::-1]
    reads[rng.random(reads.shape) < 0.001] = 4
    genome2, reads4 = orc.pack2(text), orc.pack4(reads.reshape(-1))
    gpath, rpath, opath = str(tmp_path / "g.u32"), str(tmp_path / "r.u32"), str(tmp_path / "best.bin")
    genome2.tofile(gpath); reads4.tofile(rpath)
    out = subprocess.run([EXE, "--genome", gpath, "--genome-len", str(G), "--reads", rpath, "--n-reads", str(R), "--read-len", str(M),
                          "--kmer", str(kmer), "--steps", "2", "--out", opath], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    info = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][0])
    raw = open(opath, "rb").read()
    cs = np.frombuffer(raw, dtype=np.int32, count=R)
    cp = np.frombuffer(raw, dtype=np.int64, count=R, offset=4 * R)
    crc = np.frombuffer(raw, dtype=np.uint8, count=R, offset=12 * R)

    assert info["canonical_table"] is (kmer % 2 == 1)
    fmi = amd.FMIndex.build(genome2, G, kmer_len=10, sa_int=1)       # the per-strand pass over the direct table, for both
    rb = pipeline.ReadBatch(torch.from_numpy(reads4.view(np.int32)).cuda(), R, M)
    g_dev = torch.from_numpy(genome2.view(np.int32)).cuda()
    bs, bp, brc, nc = pipeline.seed_and_extend(fmi, g_dev, G, rb, pipeline.SeedExtendParams.end_to_end())
    assert np.array_equal(cs, bs.cpu().numpy()) and np.array_equal(cp, bp.cpu().numpy()) and np.array_equal(crc, brc.cpu().numpy())
    assert info["reads"] == R and info["aligned_fraction"] > 0.99 and 0 < info["candidates"] <= nc * 1.02      # its sort + unique of a repeat's hits drops more duplicates than the adjacent compare
    fmi.close()
