#!/usr/bin/env python3
"""Round 6: the gather kernel's time follows the OUTPUT buffer's placement (tools/placement_sensitivity.py: 0.617-0.667 ms over
five 1.6-GB output buffers, the same for every table).  What about the buffer decides it?
  A  five separate allocations (addresses printed)
  B  five windows carved out of ONE 9-GB allocation at 1.7-GB steps
  C  one window of that block shifted by 4 KB / 64 KB / 2 MB / 64 MB
  D  the same five separate allocations timed again in reverse order (is a buffer's time stable?)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from scone_amd import EmbeddingCache, NGramExtractor
from scone_amd import synthetic as S

d, N, B, T = 768, 1_000_000, 2048, 512
keys, lens = S.make_keys(N, S.GPT2_VOCAB, 3, seed=11)
ex = NGramExtractor.from_arrays(keys, lens, max_n=3)
toks = [torch.from_numpy(S.stream_uniform_ids(keys, lens, B, T, 1234 + 7919 * i)).to("cuda", torch.int32) for i in range(6)]
g = torch.Generator(device="cuda").manual_seed(5)
wte = (torch.randn(S.GPT2_VOCAB, d, generator=g, device="cuda") * 0.02).half()
wpe = (torch.randn(1024, d, generator=g, device="cuda") * 0.01).half()
cache = EmbeddingCache.from_synthetic(ex, d, table_format="int8")


def run(out, n=18):
    table = cache.table
    for i in range(3):
        cache.embed_tokens(toks[i % 6], wte=wte, wpe=wpe, out=out)
    table.profile_enable(True); table.profile_read(reset=True)
    for i in range(n):
        cache.embed_tokens(toks[i % 6], wte=wte, wpe=wpe, out=out)
    k, ms = table.profile_read(reset=True)
    table.profile_enable(False)
    return ms / k


nel = B * T * d
outs = [torch.empty(B, T, d, dtype=torch.float16, device="cuda") for _ in range(5)]
print("A separate allocations:", " ".join("%#x:%.4f" % (o.data_ptr(), run(o)) for o in outs), flush=True)
big = torch.empty(9 * (1 << 30) // 2, dtype=torch.float16, device="cuda")
step = (1700 << 20) // 2
wins = [big[k * step:k * step + nel].view(B, T, d) for k in range(5)]
print("B windows of one block :", " ".join("%#x:%.4f" % (o.data_ptr(), run(o)) for o in wins), flush=True)
for sh in (0, 4 << 10, 64 << 10, 2 << 20, 64 << 20):
    o = big[sh // 2:sh // 2 + nel].view(B, T, d)
    print("C window 0 shifted by %9d B: %.4f" % (sh, run(o)), flush=True)
print("D separate, reversed   :", " ".join("%#x:%.4f" % (o.data_ptr(), run(o)) for o in reversed(outs)), flush=True)
