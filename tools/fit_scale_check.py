#!/usr/bin/env python3
"""scone_fit at scale (one-off robustness check, run on the GPU box): a 30M-token Zipf corpus in 6,000 texts;
the GPU's f-gram list (count-descending, ties in first-seen order) against numpy: same number of distinct
n-grams, same multiset of counts, and the same keys wherever a count is unique among its neighbours."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from scone_amd import synthetic as S
from scone_amd.hip_backend import fit_gpu


def main():
    n_texts, tlen, max_n = 6000, 5000, 3
    tok = S.stream_zipf(S.GPT2_VOCAB, n_texts, tlen, 21).astype(np.int64)          # [texts, tlen]
    offsets = np.arange(n_texts + 1, dtype=np.int64) * tlen
    t0 = time.time()
    keys, lens, counts, n_distinct = fit_gpu(torch.from_numpy(tok.reshape(-1)), torch.from_numpy(offsets), max_n, 2, 5_000_000)
    torch.cuda.synchronize()
    print("fit_gpu: %.2f s, %d f-grams kept, %d distinct n-grams" % (time.time() - t0, len(lens), n_distinct))
    V = np.uint64(1 << 21)
    ref_counts = []
    distinct = 0
    for n in range(1, max_n + 1):
        packed = np.zeros((n_texts, tlen - n + 1), dtype=np.uint64)
        for k in range(n):
            packed = packed * V + tok[:, k:tlen - n + 1 + k].astype(np.uint64)
        u, c = np.unique(packed.reshape(-1), return_counts=True)
        distinct += len(u)
        ref_counts.append(c[c >= 2])
        # every kept key of this length carries the count numpy finds for it
        sel = np.nonzero(lens == n)[0]
        pk = np.zeros(len(sel), dtype=np.uint64)
        for k in range(n):
            pk = pk * V + keys[sel, k].astype(np.uint64)
        pos = np.searchsorted(u, pk)
        assert np.array_equal(u[pos], pk), "a kept key does not occur in the corpus"
        assert np.array_equal(c[pos], counts[sel]), "count mismatch for length %d" % n
    ref = np.sort(np.concatenate(ref_counts))[::-1]
    assert distinct == n_distinct, (distinct, n_distinct)
    assert len(ref) >= len(lens)
    assert np.array_equal(ref[:len(lens)], counts.astype(np.int64)), "counts are not the top of the descending list"
    assert np.all(np.diff(counts.astype(np.int64)) <= 0)
    print("ok: distinct n-grams, per-key counts and the descending count list agree with numpy")


if __name__ == "__main__":
    main()
