#!/usr/bin/env python3
"""Soak: ~60 s of lookups with random batch shapes on two handles (HBM + pinned-host staged), checking that free
device memory does not drift (workspaces are re-used, nothing leaks) and that results stay bit-identical to the
first answer for every shape.  Run on the GPU box:  python tools/soak.py [seconds]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from scone_amd import EmbeddingCache, NGramExtractor
from scone_amd import synthetic as S


def main():
    secs = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
    N, d = 1_000_000, 768
    keys, lens = S.make_keys(N, S.GPT2_VOCAB, 3, seed=11)
    ex = NGramExtractor.from_arrays(keys, lens, max_n=3)
    a = EmbeddingCache.from_synthetic(ex, d, table_format="int8")
    b = EmbeddingCache.from_synthetic(ex, d, table_format="int8", placement="pinned_host", hot_rows=50257, stage_tokens=4096,
                                      cache_rows=int(os.environ.get("SCONE_SOAK_CACHE_ROWS", "0")))
    wte = (torch.randn(S.GPT2_VOCAB, d, device="cuda") * 0.02).half()
    wpe = (torch.randn(1024, d, device="cuda") * 0.01).half()
    rng = np.random.default_rng(0)
    shapes = [(1, 1), (1, 512), (3, 77), (8, 512), (64, 512), (4, 1024), (256, 512), (2, 3), (1024, 512)]
    toks = {s: torch.from_numpy(S.stream_zipf(S.GPT2_VOCAB, s[0], s[1], 7)).to("cuda", torch.int32) for s in shapes}
    if os.environ.get("SCONE_SOAK_UNIFORM") == "1":      # f-gram ids uniform over the table: the big batches reference more cold
        for s in shapes:                                   # rows than the cache holds -- eviction inside every batch
            if s[0] * s[1] >= 4096:
                toks[s] = torch.from_numpy(S.stream_uniform_ids(keys, lens, s[0], s[1], 7 + s[0])).to("cuda", torch.int32)
    first = {}
    for s in shapes:                                                     # warm every workspace size once
        first[s] = a.embed_tokens(toks[s], wte=wte, wpe=wpe).clone()
        assert torch.equal(b.embed_tokens(toks[s], wte=wte, wpe=wpe), first[s]), s
    torch.cuda.synchronize()
    def lib_bytes():       # device memory held outside torch's caching allocator = this library's tables + workspaces
        free, total = torch.cuda.mem_get_info()
        return total - free - torch.cuda.memory_reserved()
    free0 = lib_bytes()
    t0, calls, ntok = time.time(), 0, 0
    nxt = shapes[int(rng.integers(len(shapes)))]
    prefetched = 0
    while time.time() - t0 < secs:
        s, nxt = nxt, shapes[int(rng.integers(len(shapes)))]
        c = a if rng.random() < 0.5 else b
        out = c.embed_tokens(toks[s], wte=wte, wpe=wpe)
        if rng.random() < 0.4:          # round 4: the next batch's first chunks prepared ahead (used, dropped or foreign: all legal)
            b.prefetch_tokens(toks[nxt], tokens_ready=bool(rng.integers(2)))
            prefetched += 1
        calls += 1
        ntok += s[0] * s[1]
        if calls % 20 == 0:
            assert torch.equal(out, first[s]), (calls, s)
    assert a.table.status() == 0 and b.table.status() == 0
    print(f"prefetch calls {prefetched}; cache counters {b.table.stage_counters()}")
    torch.cuda.synchronize()
    free1 = lib_bytes()
    print(f"{calls} calls, {ntok / 1e9:.2f} G tokens in {time.time() - t0:.1f} s; library device memory before/after: "
          f"{free0 / 2**20:.0f} / {free1 / 2**20:.0f} MiB (torch reserved {torch.cuda.memory_reserved() / 2**20:.0f} MiB)")
    assert abs(free0 - free1) < 64 * 2**20, "device memory drifted"
    print("soak ok")


if __name__ == "__main__":
    main()
