#!/usr/bin/env python3
"""The reference's own benchmark grid (scone/configs/benchmark_config.json:79-82: batch {1, 4, 8} x
sequence {512, 1024}, 5 runs + 2 warm-up; models Small / Medium x {100K, 500K, 1M} f-grams), timed the way
scone/scripts/benchmark.py:149-200 times it (torch.cuda.synchronize + time.time around the runs), on the
MI355X path:

  lookup  = EmbeddingCache.embed_tokens (match + INT8 gather + mean + wte + wpe, one fused pass)
  forward = SconeLanguageModel.forward(input_ids) = lookup + GPT-2 body + lm_head (random-init weights of
            the named architecture, fp16, HF GPT-2 on PyTorch-ROCm -- the body is NOT part of this layer)

(The CPU baseline of the same lookup is bench.py's `cpu_baseline`; this tool does not touch oracle/.)
Run on the GPU box:  python tools/reference_grid.py [--models small,medium] [--sizes 100000,1000000]
"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from scone_amd import EmbeddingCache, NGramExtractor, SconeLanguageModel
from scone_amd import synthetic as S

ARCH = {"small": dict(n_embd=768, n_layer=12, n_head=12), "medium": dict(n_embd=1024, n_layer=24, n_head=16)}


def timed(fn, runs, warmup):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(runs):
        fn()
        torch.cuda.synchronize()
    return (time.time() - t0) * 1e3 / runs


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--models", default="small,medium")
    ap.add_argument("--sizes", default="100000,500000,1000000")
    ap.add_argument("--runs", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--no-forward", action="store_true")
    a = ap.parse_args()
    from transformers import GPT2Config, GPT2LMHeadModel
    print("| model | f-grams | B x T | lookup ms | forward ms | lookup share | lookup M tok/s |")
    print("|---|---|---|---|---|---|---|")
    for mname in a.models.split(","):
        arch = ARCH[mname]
        d = arch["n_embd"]
        model = None
        if not a.no_forward:
            torch.manual_seed(0)
            base = GPT2LMHeadModel(GPT2Config(vocab_size=S.GPT2_VOCAB, n_positions=1024, **arch)).half().cuda().eval()
        for N in (int(x) for x in a.sizes.split(",")):
            keys, lens = S.make_keys(N, S.GPT2_VOCAB, 3, seed=11)
            ex = NGramExtractor.from_arrays(keys, lens, max_n=3)
            cache = EmbeddingCache.from_synthetic(ex, d, table_format="int8")
            if not a.no_forward:
                model = SconeLanguageModel(base, embedding_cache=cache).eval()
                wte, wpe = base.transformer.wte.weight.detach(), base.transformer.wpe.weight.detach()
            else:
                wte = (torch.randn(S.GPT2_VOCAB, d, device="cuda") * 0.02).half()
                wpe = (torch.randn(1024, d, device="cuda") * 0.01).half()
            for B in (1, 4, 8):
                for T in (512, 1024):
                    tok = torch.from_numpy(S.stream_zipf(S.GPT2_VOCAB, B, T, 99)).to("cuda", torch.int32)
                    out = torch.empty(B, T, d, dtype=torch.float16, device="cuda")
                    lk = timed(lambda: cache.embed_tokens(tok, wte=wte, wpe=wpe, out=out), a.runs, a.warmup)
                    fw = float("nan")
                    if model is not None:
                        ids64 = tok.long()
                        with torch.no_grad():
                            fw = timed(lambda: model(input_ids=ids64), a.runs, a.warmup)
                    print(f"| {mname} | {N:,} | {B} x {T} | {lk:.3f} | {fw:.2f} | {lk / fw * 100 if fw == fw else float('nan'):.2f} % | "
                          f"{B * T / lk / 1e3:.1f} |", flush=True)
            del cache, model
            torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
