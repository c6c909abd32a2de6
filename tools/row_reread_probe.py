#!/usr/bin/env python3
"""How much of the gather's read traffic past L2 is RE-reads of bigram / trigram rows?  The headline table and batches, looked
up through two handles over the same rows: A = the full 1M-key index (K = 2.84 rows per token), B = an index of the 50,257
unigrams only (K = 1: wte row + unigram row + record).  Run under `rocprofv3 --pmc FETCH_SIZE --kernel-trace` (and once more
with TCC_EA0_RDREQ_sum): the k_embed_wave dispatches come in the order [warm A, warm B, N x A, N x B]; tools/runs/r05l.sh
splits the counter file by that order.  A - B = what the bigram / trigram rows cost past L2; their compulsory bytes (every
distinct row once) are printed for comparison.

    python tools/row_reread_probe.py [--steps 6]"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=6)
    a = ap.parse_args()
    import torch
    import bench
    from scone_amd import EmbeddingCache, NGramExtractor
    from scone_amd import synthetic as S
    d, B, T = 768, 2048, 512
    vocab_obj, keys, lens = bench.make_vocabulary(1_000_000, "zipf")
    full = EmbeddingCache.from_synthetic(vocab_obj, d, table_format="int8", seed=7, base_scale=0.02 / 127)
    uni = EmbeddingCache.from_synthetic(NGramExtractor.from_arrays(keys[:S.GPT2_VOCAB], lens[:S.GPT2_VOCAB], max_n=3), d,
                                        table_format="int8", seed=7, base_scale=0.02 / 127)
    g = torch.Generator(device="cuda").manual_seed(5)
    wte = (torch.randn(S.GPT2_VOCAB, d, generator=g, device="cuda") * 0.02).half()
    wpe = (torch.randn(1024, d, generator=g, device="cuda") * 0.01).half()
    _, batches = bench.make_batches(vocab_obj, keys, lens, "uniform", B, T, 1234, a.steps + 1)
    out = torch.empty(B, T, d, dtype=torch.float16, device="cuda")
    stats = {}
    for name, c in (("full", full), ("unigrams_only", uni)):
        off, ids = c.table.match_csr(batches[0])
        stats[name] = {"mean_hits_per_token": ids.numel() / (B * T), "distinct_rows": int(torch.unique(ids).numel()),
                       "distinct_rows_beyond_the_unigrams": int(torch.unique(ids[ids >= S.GPT2_VOCAB]).numel())}
        del off, ids
    torch.cuda.synchronize()
    for c in (full, uni):                      # dispatch order: warm A, warm B, N x A, N x B
        c.embed_tokens(batches[-1], wte=wte, wpe=wpe, out=out)
    torch.cuda.synchronize()
    for c in (full, uni):
        for i in range(a.steps):
            c.embed_tokens(batches[i], wte=wte, wpe=wpe, out=out)
        torch.cuda.synchronize()
    print(json.dumps({"steps": a.steps, "order": ["warm full", "warm unigrams_only", f"{a.steps} x full", f"{a.steps} x unigrams_only"],
                      "row_bytes": 770, **stats}))


if __name__ == "__main__":
    main()
