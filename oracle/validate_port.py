#!/usr/bin/env python3
"""Build-container check (needs /root/reference): the Python port in ref_port.py gives the
same outputs as the imported reference on a C1-shaped case (100K f-grams, fp32, d=768,
T=512 Zipf stream) -- asserted on all 8 sequences -- and runs at a comparable speed (it is the
`cpu_baseline` timed by bench.py).  The speed ratio is ASSERTED only loosely (0.7 .. 1.5): on
this container's 8 shared vCPUs the two loops run 5-20 % apart from one invocation to the next
(the port is usually the faster one: it skips the reference's per-call device argument
handling), so SURVEY 8d's "within +-10 %" is a typical figure, not a guarantee -- a port that is
FASTER than the reference only makes the reported CPU baseline generous to the CPU.

    python oracle/validate_port.py
"""
import os
import sys
import time
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.environ.get("SCONE_REFERENCE", "/root/reference"))
shim = types.ModuleType("scone.utils.cloud")
shim.CloudStorage = type("CloudStorage", (), {})
sys.modules["scone.utils.cloud"] = shim
from scone.tokenization.n_gram_extractor import NGramExtractor  # noqa: E402
from scone.inference.embedding_cache import EmbeddingCache      # noqa: E402
from oracle import ref_port as R                                 # noqa: E402
from scone_amd.synthetic import zipf_cdf, zipf_tokens            # noqa: E402


def main():
    torch.set_num_threads(1)
    rng = np.random.default_rng(1234)
    V, d, T = 50257, 768, 512
    cdf = zipf_cdf(V)
    corpus = [zipf_tokens(rng, cdf, 1000).tolist() for _ in range(1000)]
    ex = NGramExtractor(max_n=3, min_freq=1, max_f_grams=100_000)
    t0 = time.perf_counter(); ex.fit(corpus, verbose=False); t_fit = time.perf_counter() - t0
    n = len(ex.f_grams)
    table = torch.randn(n, d)
    cache = EmbeddingCache(ex, d)
    cache.cache_embeddings(list(range(n)), table, verbose=False)
    port = R.RefCache(dict(ex.f_gram_to_id), 3, d)
    port.embeddings = cache.embeddings
    seqs = [zipf_tokens(rng, cdf, T).tolist() for _ in range(8)]

    def reference(ids):      # engine.py:234-259, verbatim
        token_f_grams = ex.get_token_f_grams(ids)
        token_embeddings = {}
        for pos, f_grams in token_f_grams.items():
            if not f_grams:
                continue
            f_gram_ids = [ex.f_gram_to_id[f_gram] for f_gram in f_grams]
            embeddings = cache.get_embeddings(f_gram_ids, torch.device("cpu"))
            token_embeddings[pos] = embeddings.mean(dim=0)
        out = torch.zeros((1, len(ids), d), device="cpu")
        for pos, embedding in token_embeddings.items():
            out[0, pos] = embedding
        return out

    for s in seqs:                                                   # every sequence, bit for bit
        assert torch.equal(reference(s), R.aggregate(port, s, d))
    res = {}
    for name, fn in (("reference", reference), ("port", lambda s: R.aggregate(port, s, d))):
        best = 1e9
        for _ in range(3):
            t0 = time.perf_counter()
            for s in seqs:
                fn(s)
            best = min(best, time.perf_counter() - t0)
        res[name] = len(seqs) * T / best
    k = np.mean([len(v) for v in ex.get_token_f_grams(seqs[0]).values()])
    print(f"f-grams {n}, fit {t_fit:.1f} s, mean hits/token {k:.2f}")
    ratio = res["port"] / res["reference"]
    print(f"reference {res['reference']:.0f} tok/s, port {res['port']:.0f} tok/s, ratio {ratio:.3f} (1 thread); "
          f"SURVEY section 6 probed the reference at ~6,000 tok/s on this shape in a loaded container (84 ms / 512 tokens)")
    assert 0.7 < ratio < 1.5, f"the port no longer runs like the reference (ratio {ratio:.2f}): it is not a fair cpu_baseline"
    print("outputs identical on all 8 sequences; speed ratio inside (0.7, 1.5)")


if __name__ == "__main__":
    main()
