#!/usr/bin/env python3
"""Round 6: the gather kernel's time along a sustained loop.  bench.py's 20 timed steps show a transient -- the first two steps at
the trials' level (0.60 ms), a rise to 0.68 by step 6-8, a slow recovery -- in every process.  (1) 300 steps back to back, every
step's kernel time; (2) bursts of 8 steps separated by 0.2 s of idle; (3) after 2 s of idle, 60 steps."""
import os, sys, time
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
_, batches = bench.make_batches(vocab_obj, keys, lens, "uniform", B, T, 1234, 50)
out, rep = cache.alloc_output(batches[0], wte=wte, wpe=wpe, candidates=8)
table = cache.table


def loop(n):
    table.profile_enable(True); table.profile_read(reset=True)
    t0 = time.perf_counter()
    for i in range(n):
        cache.embed_tokens(batches[i % 50], wte=wte, wpe=wpe, out=out)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    s = table.profile_samples()
    table.profile_read(reset=True); table.profile_enable(False)
    return dt / n * 1e3, s


time.sleep(1.0)
ms, s = loop(300)
print("(1) 300 steps: step %.4f ms; kernel by step:" % ms)
print("    1-30  :", " ".join("%.3f" % x for x in s[:30]))
print("    every 10th from 40:", " ".join("%.3f" % x for x in s[39::10]))
for k in range(3):
    time.sleep(0.2)
    ms, s = loop(8)
    print("(2) burst of 8 after 0.2 s idle:", " ".join("%.3f" % x for x in s))
time.sleep(2.0)
ms, s = loop(60)
print("(3) 60 steps after 2 s idle: 1-20:", " ".join("%.3f" % x for x in s[:20]), " 41-60 avg %.3f" % (sum(s[40:]) / 20))
