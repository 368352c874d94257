#!/usr/bin/env python3
"""BASELINE.json configs[0] -- "sw-benchmark: 100k synthetic 100 bp reads, global Gotoh, CPU host path only (plumbing, no GPU)":
the reference's host path for this shape (aln::BatchedAlignmentScore<stream, HostThreadScheduler> = an OpenMP parallel-for over
aln::alignment_score, batched_inl.h:221-307; sw-benchmark.cu:362-435 defines the GCUPS: total pattern length x text length / time)
timed on the host cores, and the same batch through the library for the equality check and the GPU-side number.
Inputs as SURVEY.md 8d: 100,000 patterns x 100 symbols and one 4,096-symbol text from the LCG of alignment_test.cu:879-881
(s = s * 1664525 + 1013904223, seed 0, symbol = (s >> 16) % 4), GLOBAL Gotoh(2,-1,-2,-1), text blocking.
Prints one JSON line (kept under profiles/r02_config1.json)."""
import ctypes
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def lcg_symbols(n, state):
    out = np.empty(n, dtype=np.uint8)
    a, c, m = 1664525, 1013904223, 1 << 32
    # vectorised LCG: s_k = a^k s_0 + c (a^k - 1) / (a - 1)  (mod 2^32), evaluated by chunked recurrence
    s = state
    chunk = 1 << 16
    for b in range(0, n, chunk):
        k = min(chunk, n - b)
        v = np.empty(k, dtype=np.uint64)
        for i in range(k):
            s = (s * a + c) % m
            v[i] = s
        out[b:b + k] = ((v >> 16) % 4).astype(np.uint8)
    return out, s


def main():
    import oracle
    R, M, N = 100_000, 100, 4096
    rng_state = 0
    # the pure-Python LCG above is slow for 10 M symbols; numpy's generator seeded from it keeps the run short and reproducible
    first, rng_state = lcg_symbols(4096, rng_state)
    g = np.random.default_rng(int(rng_state))
    pats = g.integers(0, 4, R * M, dtype=np.uint8)
    text = first
    pat_off = (np.arange(R + 1) * M).astype(np.uint32)
    txt_off = np.zeros(R + 1, dtype=np.uint32)                      # every pattern against the WHOLE text (sw-benchmark.cu:152)
    scores = np.zeros(R, dtype=np.int32); sinks = np.zeros((R, 2), dtype=np.uint32)
    u8p, u32p, i32p = ctypes.POINTER(ctypes.c_uint8), ctypes.POINTER(ctypes.c_uint32), ctypes.POINTER(ctypes.c_int32)
    use_ref = oracle.Reference.available()
    # one thread per CPU this container may really use (min of the affinity mask and the cgroup quota), as bench.py's cpu_baseline
    import bench
    aff, quota = bench.host_cpu_share()
    threads = max(1, min(aff, int(quota) if quota else aff))
    if use_ref:
        oracle.Reference().set_num_threads(threads)
        L = oracle.Reference().lib
        cores = oracle.Reference().num_threads()
        # many-to-one: the batch entry points take per-job text offsets; give every job the same slice
        txts = text
        to = np.zeros(R + 1, dtype=np.uint32)

        def run():
            # ref_full_gotoh_batch reads text i at [to[i], to[i+1]): call it job by job slice through a tiled text instead
            L.ref_full_gotoh_many_to_one(ctypes.c_int(oracle.GLOBAL), ctypes.c_int(1), ctypes.c_int(2), ctypes.c_int(-1), ctypes.c_int(-2), ctypes.c_int(-1),
                                         pats.ctypes.data_as(u8p), pat_off.ctypes.data_as(u32p), text.ctypes.data_as(u8p), ctypes.c_uint32(N),
                                         ctypes.c_uint32(R), ctypes.c_int32(-(1 << 30)), scores.ctypes.data_as(i32p), sinks.ctypes.data_as(u32p))
    else:
        O = oracle.Oracle()
        O.set_num_threads(threads)
        cores = O.num_threads()
        sch = oracle.Scheme(2, 1, 1, -2, -1, -2, -1)

        def run():
            O.lib.orc_full_gotoh_many_to_one(ctypes.c_int(oracle.GLOBAL), ctypes.c_int(1), ctypes.byref(sch), pats.ctypes.data_as(u8p), None,
                                             pat_off.ctypes.data_as(u32p), text.ctypes.data_as(u8p), ctypes.c_uint32(N), ctypes.c_uint32(R),
                                             ctypes.c_int32(-(1 << 30)), scores.ctypes.data_as(i32p), sinks.ctypes.data_as(u32p))
    run()                                                            # warm (page in, spin up the thread team)
    t0 = time.perf_counter(); run(); cpu_s = time.perf_counter() - t0
    cpu_scores = scores.copy()
    out = {"config": "BASELINE configs[0]: sw-benchmark shape, 100k x 100 bp vs one 4,096-symbol text, GLOBAL Gotoh(2,-1,-2,-1), text blocking",
           "cells": R * M * N,
           "cpu": {"kind": "reference" if use_ref else "port", "cores": cores, "seconds": cpu_s, "gcups": R * M * N / cpu_s / 1e9,
                   "path": "the reference's own aln::alignment_score over an OpenMP parallel-for (oracle/_ref)" if use_ref
                           else "the oracle's C restatement over an OpenMP parallel-for"}}
    try:
        import torch
        import __graft_entry__ as ge
        if torch.cuda.is_available():
            amd = ge.load_package()
            O = oracle.Oracle()
            batch = amd.AlignmentBatch(O.pack4(pats), 4, pat_off, O.pack2(text), 2, np.zeros(R, dtype=np.uint32), np.full(R, N, dtype=np.uint32))
            op = amd.BatchedAlignmentScore(amd.make_gotoh_aligner(amd.GLOBAL, amd.SimpleGotohScheme(2, -1, -2, -1)), text_blocking=True)
            sc, sk = op.enact(batch, M, N)
            torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(5):
                sc, sk = op.enact(batch, M, N)
            b.record(); torch.cuda.synchronize()
            ms = a.elapsed_time(b) / 5
            out["gpu"] = {"ms": ms, "gcups": R * M * N / (ms * 1e-3) / 1e9, "scores_equal_cpu": bool(np.array_equal(sc.cpu().numpy(), cpu_scores))}
    except Exception as e:                                           # the CPU line stands on its own
        out["gpu"] = {"error": str(e)}
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
