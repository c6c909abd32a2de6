#!/usr/bin/env python3
"""What stock PyTorch-ROCm ops give for the same step on the same GPU (context, not a target): C2's table (1M rows fp16,
d = 768) as a torch tensor, the per-token id lists taken from this repo's match (so only the gather / mean / combine is torch's:
`F.embedding_bag(mode="mean")` over the CSR lists + `wte[tok]` + `wpe`), against `scone_embed` (match INCLUDED) on the same
batch.  Prints one JSON object.     python tools/torch_baseline.py [--steps 20]"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=20)
    a = ap.parse_args()
    import torch
    import torch.nn.functional as F
    import bench
    from scone_amd import EmbeddingCache
    from scone_amd import synthetic as S
    d, B, T = 768, 2048, 512
    vocab_obj, keys, lens = bench.make_vocabulary(1_000_000, "zipf")
    cache = EmbeddingCache.from_synthetic(vocab_obj, d, table_format="fp16", seed=7, base_scale=0.02 / 127)
    g = torch.Generator(device="cuda").manual_seed(5)
    wte = (torch.randn(S.GPT2_VOCAB, d, generator=g, device="cuda") * 0.02).half()
    wpe = (torch.randn(1024, d, generator=g, device="cuda") * 0.01).half()
    _, batches = bench.make_batches(vocab_obj, keys, lens, "uniform", B, T, 1234, a.steps)
    # the fp16 table as a plain tensor (every row through the library's own row gather: the same values)
    table = torch.empty(1_000_000, d, dtype=torch.float16, device="cuda")
    for r0 in range(0, 1_000_000, 100_000):
        table[r0:r0 + 100_000] = cache.table.gather_rows(torch.arange(r0, r0 + 100_000, device="cuda")).half()
    lists = []
    for t in batches:
        off, ids = cache.table.match_csr(t)
        lists.append((off[:-1].to(torch.int64).contiguous(), ids.to(torch.int64).contiguous()))
    pos = torch.arange(T, device="cuda")

    def torch_step(i):
        off, ids = lists[i]
        fg = F.embedding_bag(ids, table, off, mode="mean")                      # [B*T, d] fp16 (fp32 accumulate inside)
        return (wte[batches[i].view(-1).long()] + fg).view(B, T, d) + wpe[pos]   # language_model.py:239-254

    out = torch.empty(B, T, d, dtype=torch.float16, device="cuda")

    def scone_step(i):
        return cache.embed_tokens(batches[i], wte=wte, wpe=wpe, out=out)
    res = {}
    for name, fn in (("torch_embedding_bag", torch_step), ("scone_embed", scone_step)):
        for i in range(3):
            fn(i)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(a.steps):
            fn(i)
        torch.cuda.synchronize()
        res[name] = {"ms_per_step": (time.perf_counter() - t0) / a.steps * 1e3}
        res[name]["tokens_per_s"] = B * T / res[name]["ms_per_step"] * 1e3
    ref = torch_step(0).float()
    got = scone_step(0).float()
    res["max_rel_diff"] = float((ref - got).abs().max() / ref.abs().max())
    res["what"] = __doc__.split("\n\n")[0]
    res["torch"] = torch.__version__
    res["speedup"] = res["torch_embedding_bag"]["ms_per_step"] / res["scone_embed"]["ms_per_step"]
    print(json.dumps(res))


if __name__ == "__main__":
    main()
