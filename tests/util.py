"""shared helpers for the parity tests"""
import numpy as np


def make_queries(rng, text, Q, min_len=1, max_len=32, hit_every=2, n_count=0):
    """concatenated query set: every `hit_every`-th query is a substring of text, others random"""
    n = len(text)
    lens = rng.integers(min_len, max_len + 1, Q)
    offs = np.zeros(Q + 1, dtype=np.uint32)
    offs[1:] = np.cumsum(lens)
    syms = rng.integers(0, 4, int(offs[-1]), dtype=np.uint8)
    for q in range(0, Q, hit_every):
        if lens[q] <= n:
            p = int(rng.integers(0, n - lens[q] + 1))
            syms[offs[q]:offs[q + 1]] = text[p:p + lens[q]]
    if n_count:
        syms[rng.integers(0, len(syms), n_count)] = 4
    return syms, offs


def mutate_reads(rng, text, starts, M, sub=0.02, indel=0.1):
    """reads of length M drawn at `starts`, with substitutions and an occasional 1-3 bp indel"""
    reads = np.zeros((len(starts), M), dtype=np.uint8)
    for k, s in enumerate(starts):
        r = text[s:s + M + 8].copy()
        if rng.random() < indel:
            p = int(rng.integers(5, M - 5)); g = int(rng.integers(1, 4))
            if rng.random() < 0.5:
                r = np.concatenate([r[:p], r[p + g:]])
            else:
                r = np.concatenate([r[:p], rng.integers(0, 4, g, dtype=np.uint8), r[p:]])
        r = r[:M]
        m = rng.random(M) < sub
        r[m] = rng.integers(0, 4, int(m.sum()))
        reads[k] = r
    return reads
