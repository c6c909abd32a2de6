#!/usr/bin/env python3
"""Latency -> bandwidth regime of the fused lookup: tokens per launch in
{512, 4096, 65536, 1M}, both streams, per-kernel times from HIP events.  Run on the GPU box:
    python tools/sweep.py
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from scone_amd import EmbeddingCache, NGramExtractor
from scone_amd import synthetic as S
from scone_amd.hip_backend import format_code, row_bytes


def main():
    d, N = 768, 1_000_000
    keys, lens = S.make_keys(N)
    cache = EmbeddingCache.from_synthetic(NGramExtractor.from_arrays(keys, lens, max_n=3), d, table_format="int8")
    table = cache.table
    wte = (torch.randn(S.GPT2_VOCAB, d, device="cuda") * 0.02).half()
    wpe = (torch.randn(1024, d, device="cuda") * 0.01).half()
    print("| stream | B x T | tokens | K mean | step us | gather kernel us | M tokens/s | algorithmic GB/s |")
    print("|---|---|---|---|---|---|---|---|")
    for stream in ("uniform", "zipf"):
        for B, T in ((1, 512), (8, 512), (128, 512), (2048, 512)):
            tok_np = S.stream_uniform_ids(keys, lens, B, T, 5) if stream == "uniform" else S.stream_zipf(S.GPT2_VOCAB, B, T, 5)
            tok = torch.from_numpy(tok_np).to("cuda", torch.int32)
            out = torch.empty(B, T, d, dtype=torch.float16, device="cuda")
            off, _ = table.match_csr(tok)
            sum_k = int(off[-1].item())
            nbytes = sum_k * row_bytes(format_code("int8"), d) + B * T * (d * 4 + 4)
            steps = 200 if B * T < 100_000 else 30
            for _ in range(10):
                cache.embed_tokens(tok, wte=wte, wpe=wpe, out=out)
            table.profile_enable(True)
            table.profile_read(reset=True)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(steps):
                cache.embed_tokens(tok, wte=wte, wpe=wpe, out=out)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / steps
            n, ms = table.profile_read(reset=True)
            table.profile_enable(False)
            print(f"| {stream} | {B} x {T} | {B * T} | {sum_k / (B * T):.2f} | {dt * 1e6:.1f} | {ms / n * 1e3:.1f} | "
                  f"{B * T / dt / 1e6:.1f} | {nbytes / (ms / n * 1e-3) / 1e9:.0f} |")


if __name__ == "__main__":
    main()
