#!/usr/bin/env python3
"""Headline step with the token ids starting on the HOST (pinned memory): H2D copy of the [2048, 512] int32 ids + the lookup,
per step; printed next to the resident-input rate.  (the PCIe-inclusive rate is a note, never `value`.)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from scone_amd import EmbeddingCache, NGramExtractor
from scone_amd import synthetic as S
d, N, B, T = 768, 1_000_000, 2048, 512
keys, lens = S.make_keys(N, S.GPT2_VOCAB, 3, seed=11)
cache = EmbeddingCache.from_synthetic(NGramExtractor.from_arrays(keys, lens, max_n=3), d, table_format="int8")
tok_h = torch.from_numpy(S.stream_uniform_ids(keys, lens, B, T, 1234)).to(torch.int32).pin_memory()
tok_d = tok_h.cuda()
wte = (torch.randn(S.GPT2_VOCAB, d, device="cuda") * 0.02).half(); wpe = (torch.randn(1024, d, device="cuda") * 0.01).half()
out = torch.empty(B, T, d, dtype=torch.float16, device="cuda")
def run(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n
a = run(lambda: cache.embed_tokens(tok_d, wte=wte, wpe=wpe, out=out))
b = run(lambda: cache.embed_tokens(tok_h.to("cuda", non_blocking=True), wte=wte, wpe=wpe, out=out))
print(f"ids resident in HBM: {a * 1e3:.3f} ms/step, {B * T / a / 1e9:.3f} G tok/s;  ids from pinned host memory: {b * 1e3:.3f} ms/step, {B * T / b / 1e9:.3f} G tok/s")
