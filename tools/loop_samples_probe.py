#!/usr/bin/env python3
"""Round 6: per-step kernel times of bench.py's timed loop (5 warm-up + 20 timed steps, a different batch every step), in the
order bench.py does things -- output buffer chosen, THEN the workload statistics (match_csr, unique: ~0.1 s of mostly idle GPU),
then the loop -- and with the statistics first, so that the trials of alloc_output run right before the loop.
    python tools/loop_samples_probe.py [stats_last|stats_first]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from scone_amd import EmbeddingCache
from scone_amd import synthetic as S
from scone_amd.hip_backend import format_code

order = sys.argv[1] if len(sys.argv) > 1 else "stats_last"
d, B, T = 768, 2048, 512
vocab_obj, keys, lens = bench.make_vocabulary(1_000_000, "zipf")
cache = EmbeddingCache.from_synthetic(vocab_obj, d, table_format="int8", seed=7, base_scale=0.02 / 127)
g = torch.Generator(device="cuda").manual_seed(5)
wte = (torch.randn(S.GPT2_VOCAB, d, generator=g, device="cuda") * 0.02).half()
wpe = (torch.randn(1024, d, generator=g, device="cuda") * 0.01).half()
_, batches = bench.make_batches(vocab_obj, keys, lens, "uniform", B, T, 1234, 25)
if order == "stats_first":
    bench.workload_bytes(cache.table, batches[0], format_code("int8"), d)
out, rep = cache.alloc_output(batches[0], wte=wte, wpe=wpe, candidates=8)
if order == "stats_last":
    bench.workload_bytes(cache.table, batches[0], format_code("int8"), d)
dt, n_launch, kern_ms, samples = bench.lookup_loop(cache, batches, wte, wpe, out, 20, 5, torch.cuda.synchronize, False)
print(order, "finalists", {k: round(v, 4) for k, v in rep["finalists_kernel_ms"].items()}, "kept", rep["kept"])
print(order, "step %.4f ms  kernel avg %.4f" % (dt / 20 * 1e3, kern_ms / n_launch), "samples", " ".join("%.3f" % s for s in samples), flush=True)
