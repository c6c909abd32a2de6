#!/usr/bin/env python3
"""Which allocation decides the gather kernel's mode?  One process: several tables kept alive (different physical
placements) x one out buffer, then one table x several out buffers kept alive."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from scone_amd import EmbeddingCache, NGramExtractor
from scone_amd import synthetic as S

d, N, B, T = 768, 1_000_000, 2048, 512
keys, lens = S.make_keys(N, S.GPT2_VOCAB, 3, seed=11)
ex = NGramExtractor.from_arrays(keys, lens, max_n=3)
tok = torch.from_numpy(S.stream_uniform_ids(keys, lens, B, T, 1234)).to("cuda", torch.int32)
g = torch.Generator(device="cuda").manual_seed(5)
wte = (torch.randn(S.GPT2_VOCAB, d, generator=g, device="cuda") * 0.02).half()
wpe = (torch.randn(1024, d, generator=g, device="cuda") * 0.01).half()


def run(cache, out):
    table = cache.table
    for _ in range(5):
        cache.embed_tokens(tok, wte=wte, wpe=wpe, out=out)
    table.profile_enable(True); table.profile_read(reset=True)
    for _ in range(30):
        cache.embed_tokens(tok, wte=wte, wpe=wpe, out=out)
    n, ms = table.profile_read(reset=True)
    table.profile_enable(False)
    return ms / n


caches = [EmbeddingCache.from_synthetic(ex, d, table_format="int8") for _ in range(5)]
outs = [torch.empty(B, T, d, dtype=torch.float16, device="cuda") for _ in range(5)]
print("full matrix (rows: tables, columns: out buffers), ms per gather kernel:", flush=True)
for i, c in enumerate(caches):
    print("  table[%d]:" % i, " ".join("%.4f" % run(c, o) for o in outs), flush=True)
print("tables x out[0]:", " ".join("%.4f" % run(c, outs[0]) for c in caches), flush=True)
print("table[0] x outs:", " ".join("%.4f" % run(caches[0], o) for o in outs), flush=True)
wtes = [wte.clone() for _ in range(4)]
res = []
for w in wtes:
    wte = w
    res.append(run(caches[0], outs[0]))
print("table[0] x out[0] x wtes:", " ".join("%.4f" % r for r in res), flush=True)
