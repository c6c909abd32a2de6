#!/usr/bin/env python3
"""What in a process's stream state slows the staging pipeline of a pinned-host table (round 6: bench.py's `latency` block earlier
in the process cost the cached step 27 %, profiles/r06d)?  One condition per process, then the C4 table behind its 16M-slot cache,
Zipf-ids stream, a different batch every step with scone_embed_prefetch (bench.py's n1_pinned_host_zipf loop):

    none      nothing before the table is built
    streams   six torch side streams, one small kernel on each
    graph     one hipGraph capture + replay of a small torch kernel on a side stream (no scone call inside)
    lookups   the latency block's eager part only (small lookups on the default stream)
    latency   the whole latency block (bench.latency_block)

    python tools/stream_state_probe.py <condition> [--rows 100000000] [--warm 150] [--steps 20]"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("condition", choices=["none", "streams", "graph", "lookups", "latency"])
    ap.add_argument("--rows", type=int, default=100_000_000)
    ap.add_argument("--warm", type=int, default=150)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--n-streams", type=int, default=6, help="condition `streams`: how many torch side streams are used before the table is built")
    a = ap.parse_args()
    import torch
    import bench
    from scone_amd import EmbeddingCache
    from scone_amd import synthetic as S
    torch.cuda.init()
    x = torch.zeros(1 << 20, device="cuda")
    if a.condition == "streams":
        keep = []
        for _ in range(a.n_streams):
            s = torch.cuda.Stream()
            with torch.cuda.stream(s):
                x.add_(1.0)
            keep.append(s)
        torch.cuda.synchronize()
    elif a.condition == "graph":
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            x.add_(1.0)
        torch.cuda.current_stream().wait_stream(side)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=side):
            x.add_(1.0)
        for _ in range(10):
            g.replay()
        torch.cuda.synchronize()
        del g
    elif a.condition in ("lookups", "latency"):
        d0 = 768
        vocab_obj, keys, lens = bench.make_vocabulary(1_000_000, "zipf")
        c0 = EmbeddingCache.from_synthetic(vocab_obj, d0, table_format="int8", seed=7, base_scale=0.02 / 127)
        g0 = torch.Generator(device="cuda").manual_seed(5)
        wte0 = (torch.randn(S.GPT2_VOCAB, d0, generator=g0, device="cuda") * 0.02).half()
        wpe0 = (torch.randn(1024, d0, generator=g0, device="cuda") * 0.01).half()
        if a.condition == "latency":
            bench.latency_block(c0, wte0, wpe0, d0, calls=100)
        else:
            for B, T in bench.REFERENCE_GRID:
                tok = torch.from_numpy(S.stream_zipf(S.GPT2_VOCAB, B, T, 99)).to("cuda", torch.int32)
                for _ in range(100):
                    c0.embed_tokens(tok, wte=wte0, wpe=wpe0)
            torch.cuda.synchronize()
        del c0, wte0, wpe0
        torch.cuda.empty_cache()
    N, d, B, T = a.rows, 1024, 2048, 512
    hot = min(1_000_000, max(N // 100, 1))
    vocab = S.StructuredVocab(N)
    g = torch.Generator(device="cuda").manual_seed(5)
    wte = (torch.randn(S.GPT2_VOCAB, d, generator=g, device="cuda") * 0.02).half()
    wpe = (torch.randn(1024, d, generator=g, device="cuda") * 0.01).half()
    out = torch.empty(B, T, d, dtype=torch.float16, device="cuda")
    cache = EmbeddingCache.from_synthetic(vocab, d, table_format="int4", seed=7, base_scale=0.02 / 127, n_rows=N, placement="pinned_host",
                                          hot_rows=hot, stage_tokens=262144, cache_rows=min(16_000_000, N - hot))
    cache.table.reserve(B * T)
    for i in range(a.warm):
        cache.embed_tokens(S.stream_zipf_ids_torch(vocab, B, T, 1234 + i), wte=wte, wpe=wpe, out=out)
    timed = [S.stream_zipf_ids_torch(vocab, B, T, 50_000 + i) for i in range(a.steps)]
    torch.cuda.synchronize()
    c0 = cache.table.stage_counters()
    t0 = time.perf_counter()
    for i in range(a.steps):
        cache.embed_tokens(timed[i], wte=wte, wpe=wpe, out=out)
        if i + 1 < a.steps:
            cache.prefetch_tokens(timed[i + 1], tokens_ready=True)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / a.steps
    c1 = cache.table.stage_counters()
    print(json.dumps({"condition": a.condition, "ms_per_step": dt * 1e3, "G_tokens_per_s": B * T / dt / 1e9,
                      "n_streams": a.n_streams if a.condition == "streams" else None,
                      "rows_over_pcie_per_step": (c1["rows_copied"] - c0["rows_copied"]) / a.steps,
                      "GPU_MAX_HW_QUEUES": os.environ.get("GPU_MAX_HW_QUEUES")}), flush=True)


if __name__ == "__main__":
    main()
