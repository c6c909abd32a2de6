#!/usr/bin/env python3
"""BASELINE config C4 (100M rows INT4 d = 1024 in pinned host DRAM, first `--hot` rows in HBM) on the realistic stream --
f-gram ids drawn from a power law over the frequency-ordered table -- with a DIFFERENT batch every step, which is what a
cache of cold rows must be measured on (bench.py's round-3 figure re-used one batch: fine for zero-copy and for a per-chunk
staging buffer, which forget everything between steps, meaningless for anything that keeps rows across steps).

One mechanism per process (`--mode`), so that it can sit directly behind `rocprofv3 ... --`:
  zero      the lookup kernel reads cold rows in place over PCIe
  staged    per-chunk staging buffer (`--stage-tokens`), round-1..3 form: nothing survives a chunk
  cached    persistent HBM cache of cold rows (`--cache-rows`) in front of the same staged pipeline
Prints one JSON line: ms/step (mean of the timed steps, and per step), cold references / distinct cold rows per step, and --
cached -- the rows that crossed PCIe per step (the library's counters).
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from scone_amd import EmbeddingCache
from scone_amd import synthetic as S


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=100_000_000)
    ap.add_argument("--hot", type=int, default=1_000_000)
    ap.add_argument("--mode", default="zero", choices=["zero", "staged", "cached"])
    ap.add_argument("--stage-tokens", type=int, default=262144)
    ap.add_argument("--cache-rows", type=int, default=8_000_000)
    ap.add_argument("--batch", type=int, default=2048)
    ap.add_argument("--seq", type=int, default=512)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--same-batch", action="store_true", help="every step the same batch (round 3's measurement)")
    ap.add_argument("--prefetch-next", action="store_true", help="cached: scone_embed_prefetch of batch i + 1 right after the "
                    "lookup of batch i is queued (a loop that knows its next tokens early)")
    ap.add_argument("--scramble", action="store_true", help="popularity rank r is served by row (r * M + shift) %% N: the table's "
                    "order is not the traffic's frequency order (synthetic.stream_zipf_ids_torch)")
    ap.add_argument("--shift-per-step", type=int, default=0, help="the hot set moves by this many rows from batch to batch")
    ap.add_argument("--cu-reserve", type=int, default=0, help="scone_set_cu_reserve: the lookup kernels leave this many CUs to the "
                    "cache's copy and preparation kernels; the loop runs on the handle's masked stream")
    ap.add_argument("--stats-steps", type=int, default=2, help="steps whose cold references are counted (host sync: not timed)")
    a = ap.parse_args()
    N, d, B, T = a.rows, 1024, a.batch, a.seq
    vocab = S.StructuredVocab(N)
    kw = {}
    if a.mode in ("staged", "cached"):
        kw["stage_tokens"] = a.stage_tokens
    if a.mode == "cached":
        kw["cache_rows"] = a.cache_rows
    t0 = time.perf_counter()
    cache = EmbeddingCache.from_synthetic(vocab, d, table_format="int4", seed=7, base_scale=0.02 / 127, n_rows=N,
                                          placement="pinned_host", hot_rows=a.hot, **kw)
    torch.cuda.synchronize()
    build_s = time.perf_counter() - t0
    g = torch.Generator(device="cuda").manual_seed(5)
    wte = (torch.randn(S.GPT2_VOCAB, d, generator=g, device="cuda") * 0.02).half()
    wpe = (torch.randn(1024, d, generator=g, device="cuda") * 0.01).half()
    out = torch.empty(B, T, d, dtype=torch.float16, device="cuda")
    n_batches = 1 if a.same_batch else a.steps + a.warmup
    toks = [S.stream_zipf_ids_torch(vocab, B, T, 1234 + i, scramble=a.scramble, shift=i * a.shift_per_step) for i in range(n_batches)]
    torch.cuda.synchronize()
    stats = []
    for i in range(min(a.stats_steps, n_batches)):
        _, ids = cache.table.match_csr(toks[(a.warmup + i) % n_batches])            # of the TIMED batches
        cold = ids[ids >= a.hot]
        stats.append({"cold_row_references": int(cold.numel()), "distinct_cold_rows": int(torch.unique(cold).numel()),
                      "mean_hits_per_token": float(ids.numel()) / (B * T)})
        del ids, cold
    cache.table.reserve(B * T)
    import contextlib
    ctx = contextlib.nullcontext()
    if a.cu_reserve:
        cache.table.set_cu_reserve(a.cu_reserve)
        ctx = torch.cuda.stream(cache.table.lookup_stream())
        torch.cuda.synchronize()
    with ctx:
        timed_loop(a, cache, toks, n_batches, wte, wpe, out, B, T, N, build_s, stats)


def timed_loop(a, cache, toks, n_batches, wte, wpe, out, B, T, N, build_s, stats):
    for i in range(a.warmup):
        cache.embed_tokens(toks[i % n_batches], wte=wte, wpe=wpe, out=out)
    torch.cuda.synchronize()
    c0 = cache.table.stage_counters() if hasattr(cache.table, "stage_counters") else None
    per = []
    t_all = time.perf_counter()
    for i in range(a.steps):
        cache.embed_tokens(toks[(a.warmup + i) % n_batches], wte=wte, wpe=wpe, out=out)
        if a.prefetch_next and i + 1 < a.steps:
            cache.prefetch_tokens(toks[(a.warmup + i + 1) % n_batches], tokens_ready=True)   # (all batches were generated up front)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t_all) / a.steps
    for i in range(min(a.steps, 8)):                     # a few steps one by one (host-synchronised: upper bound per step)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        cache.embed_tokens(toks[(a.warmup + i) % n_batches], wte=wte, wpe=wpe, out=out)
        torch.cuda.synchronize()
        per.append((time.perf_counter() - t1) * 1e3)
    res = {"mode": a.mode, "rows": N, "hot_rows": a.hot, "tokens": B * T, "build_s": build_s, "steps": a.steps, "warmup": a.warmup,
           "different_batch_every_step": not a.same_batch, "scramble": a.scramble, "shift_per_step": a.shift_per_step, "prefetch_next": a.prefetch_next, "cu_reserve": a.cu_reserve, "ms_per_step": dt * 1e3, "tokens_per_s": B * T / dt,
           "ms_single_steps": [round(x, 4) for x in per], "per_batch_stats": stats, "status": cache.table.status(),
           "checksum_last": float(out.float().abs().sum().item())}
    if a.mode != "zero":
        res["stage_tokens"] = a.stage_tokens
    if a.mode == "cached":
        res["cache_rows"] = a.cache_rows
    if c0 is not None:
        c1 = cache.table.stage_counters()
        n = a.steps + min(a.steps, 8)
        res["per_step"] = {k: (c1[k] - c0[k]) / n for k in c1}
    print(json.dumps(res))


if __name__ == "__main__":
    main()
