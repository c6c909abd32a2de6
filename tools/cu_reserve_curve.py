#!/usr/bin/env python3
"""What a CU reserve costs the lookup kernel (scone_set_cu_reserve): the headline step (INT8 1M x 768, S_uniform, 2048 x 512
tokens) and the C4-in-HBM step (INT4 100M x 1024) with R = 0, 8, 16, 32, 64 compute units left free, alternating in one
process (medians of --rounds).  One JSON line."""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from scone_amd import EmbeddingCache, NGramExtractor
from scone_amd import synthetic as S


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--reserves", default="0,8,16,32,64")
    ap.add_argument("--configs", default="headline,c4_hbm")
    ap.add_argument("--full-mask", action="store_true", help="SCONE_CU_RESERVE_DEBUG_FULL_MASK=1: the hop to the handle's stream "
                    "with every CU enabled (what the two events cost, apart from the masking)")
    a = ap.parse_args()
    if a.full_mask:
        os.environ["SCONE_CU_RESERVE_DEBUG_FULL_MASK"] = "1"
    B, T = 2048, 512
    res = {}
    for cfg in a.configs.split(","):
        if cfg == "headline":
            keys, lens = S.make_keys(1_000_000, S.GPT2_VOCAB, 3, seed=11)
            ex = NGramExtractor.from_arrays(keys, lens, max_n=3)
            d, fmt = 768, "int8"
            cache = EmbeddingCache.from_synthetic(ex, d, table_format=fmt, seed=7, base_scale=0.02 / 127)
            tok = torch.from_numpy(S.stream_uniform_ids(keys, lens, B, T, 1234)).to("cuda", torch.int32)
        else:
            vocab = S.StructuredVocab(100_000_000)
            d, fmt = 1024, "int4"
            cache = EmbeddingCache.from_synthetic(vocab, d, table_format=fmt, seed=7, base_scale=0.02 / 127, n_rows=100_000_000)
            tok = torch.from_numpy(S.stream_uniform_ids(vocab, None, B, T, 1234)).to("cuda", torch.int32)
        g = torch.Generator(device="cuda").manual_seed(5)
        wte = (torch.randn(S.GPT2_VOCAB, d, generator=g, device="cuda") * 0.02).half()
        wpe = (torch.randn(1024, d, generator=g, device="cuda") * 0.01).half()
        out = torch.empty(B, T, d, dtype=torch.float16, device="cuda")
        t = cache.table
        t.reserve(B * T)
        rs = [int(x) for x in a.reserves.split(",")]
        samples = {r: [] for r in rs}
        kern = {r: [] for r in rs}
        sums = {}
        for r in rs:
            t.set_cu_reserve(r)
            cache.embed_tokens(tok, wte=wte, wpe=wpe, out=out)
            torch.cuda.synchronize()
            sums[r] = float(out.float().abs().sum().item())
        for _ in range(a.rounds):
            for r in rs:
                t.set_cu_reserve(r)
                cache.embed_tokens(tok, wte=wte, wpe=wpe, out=out)
                torch.cuda.synchronize()
                t.profile_enable(True)
                t.profile_read(reset=True)
                t0 = time.perf_counter()
                for _ in range(a.steps):
                    cache.embed_tokens(tok, wte=wte, wpe=wpe, out=out)
                torch.cuda.synchronize()
                samples[r].append((time.perf_counter() - t0) * 1e3 / a.steps)
                n, ms = t.profile_read(reset=True)
                t.profile_enable(False)
                kern[r].append(ms / max(n, 1))
        t.set_cu_reserve(0)
        med = lambda v: sorted(v)[len(v) // 2]
        res[cfg] = {"step_ms": {r: round(med(samples[r]), 4) for r in rs},
                    "lookup_kernel_ms_incl_event_hop": {r: round(med(kern[r]), 4) for r in rs},
                    "same_output": len(set(sums.values())) == 1, "status": t.status(), "cus": t.cu_reserve()[1]}
        del cache, t, tok, out, wte, wpe
        torch.cuda.empty_cache()
    print(json.dumps({"what": "lookup step vs compute units reserved (masked stream)" + (" -- DEBUG: every CU enabled, hop only" if a.full_mask else ""), "tokens": B * T, "steps": a.steps,
                      "rounds": a.rounds, "configs": res}))


if __name__ == "__main__":
    main()
