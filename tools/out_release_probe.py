#!/usr/bin/env python3
"""Round 6: alloc_output's trials (one batch repeated, all candidates alive) run the chosen buffer ~6 % faster than the timed loop
does afterwards.  Which difference is it?  (a) all candidates alive, batch repeated; (b) the same after the other candidates
were released; (c) alive again (8 fresh ones allocated), batch repeated; (d) rotating batches; (e) repeated batch again."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from scone_amd import EmbeddingCache
from scone_amd import synthetic as S

d, B, T = 768, 2048, 512
vocab_obj, keys, lens = bench.make_vocabulary(1_000_000, "zipf")
cache = EmbeddingCache.from_synthetic(vocab_obj, d, table_format="int8", seed=7, base_scale=0.02 / 127)
g = torch.Generator(device="cuda").manual_seed(5)
wte = (torch.randn(S.GPT2_VOCAB, d, generator=g, device="cuda") * 0.02).half()
wpe = (torch.randn(1024, d, generator=g, device="cuda") * 0.01).half()
_, batches = bench.make_batches(vocab_obj, keys, lens, "uniform", B, T, 1234, 25)
table = cache.table


def run(out, toks, n=20):
    for i in range(3):
        cache.embed_tokens(toks[i % len(toks)], wte=wte, wpe=wpe, out=out)
    table.profile_enable(True); table.profile_read(reset=True)
    for i in range(n):
        cache.embed_tokens(toks[i % len(toks)], wte=wte, wpe=wpe, out=out)
    k, ms = table.profile_read(reset=True)
    table.profile_enable(False)
    return ms / k


bufs = [torch.empty(B, T, d, dtype=torch.float16, device="cuda") for _ in range(8)]
for o in bufs:
    cache.embed_tokens(batches[0], wte=wte, wpe=wpe, out=o)
torch.cuda.synchronize()
t = [run(o, batches[:1], 5) for o in bufs]
kept = min(range(8), key=lambda i: t[i])
out = bufs[kept]
print("trials (all alive, one batch):", " ".join("%.4f" % x for x in t), "kept", kept, flush=True)
print("(a) kept, all alive, one batch repeated x20 : %.4f" % run(out, batches[:1]), flush=True)
print("(a') kept, all alive, 25 batches rotating   : %.4f" % run(out, batches), flush=True)
del bufs, o
torch.cuda.empty_cache()
print("(b) kept, others released, one batch        : %.4f" % run(out, batches[:1]), flush=True)
print("(d) kept, others released, rotating         : %.4f" % run(out, batches), flush=True)
ballast = [torch.empty(B, T, d, dtype=torch.float16, device="cuda") for _ in range(7)]
print("(c) kept, 7 fresh buffers alive, one batch  : %.4f" % run(out, batches[:1]), flush=True)
print("(c') kept, 7 fresh buffers alive, rotating  : %.4f" % run(out, batches), flush=True)
del ballast
torch.cuda.empty_cache()
print("(e) kept, released again, one batch         : %.4f" % run(out, batches[:1]), flush=True)
