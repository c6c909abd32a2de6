#!/usr/bin/env python3
"""In-process A/B of the serving loop with and without scone_embed_prefetch (the next batch's match on the handle's side
stream), per batch size: process-to-process placement moves the headline kernel by 5-10 %, so the two forms alternate on
ONE table and ONE set of batches.  Headline table (1M-row INT8, d = 768), S_uniform stream, a different batch every step.

    python tools/prefetch_sweep.py [--sizes 64,128,256,512,1024,2048] [--steps 30] [--rounds 3]

NOTE: the announced match was removed after this measurement (profiles/r05c: 1-19 % slower at every size); on the current
tree scone_embed_prefetch is a no-op for HBM tables and both loops run the same thing.  To repeat the experiment apply
profiles/r05b/announced_match_implementation.diff first.

Prints one JSON object: per batch size (sequences of 512 tokens) the median over rounds of ms/step and of the gather kernel's
HIP-event time for both loops."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--sizes", default="64,128,256,512,1024,2048")
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--format", default="int8")
    ap.add_argument("--dim", type=int, default=768)
    ap.add_argument("--rows", type=int, default=1_000_000)
    a = ap.parse_args()
    import numpy as np
    import torch
    import bench
    from scone_amd import EmbeddingCache
    from scone_amd import synthetic as S
    vocab_obj, keys, lens = bench.make_vocabulary(a.rows, "zipf")
    cache = EmbeddingCache.from_synthetic(vocab_obj, a.dim, table_format=a.format, seed=7, base_scale=0.02 / 127)
    g = torch.Generator(device="cuda").manual_seed(5)
    wte = (torch.randn(S.GPT2_VOCAB, a.dim, generator=g, device="cuda") * 0.02).half()
    wpe = (torch.randn(1024, a.dim, generator=g, device="cuda") * 0.01).half()
    T = 512
    res = {"what": __doc__.split("\n\n")[0], "table": f"{a.rows}-row {a.format} d={a.dim}", "steps": a.steps, "rounds": a.rounds, "sizes": {}}
    for B in [int(x) for x in a.sizes.split(",")]:
        _, batches = bench.make_batches(vocab_obj, keys, lens, "uniform", B, T, 1234, min(a.steps + 4, 40))
        out = torch.empty(B, T, a.dim, dtype=torch.float16, device="cuda")
        ref = cache.embed_tokens(batches[0], wte=wte, wpe=wpe).clone()
        rows = {"serial": [], "announced": []}
        for _ in range(a.rounds):
            for name, pf in (("serial", False), ("announced", True)):
                dt, nl, km, sm = bench.lookup_loop(cache, batches, wte, wpe, out, a.steps, 4, torch.cuda.synchronize, pf)
                rows[name].append((dt / a.steps * 1e3, km / max(nl, 1)))
        cache.prefetch_tokens(batches[0], tokens_ready=True)
        same = bool(torch.equal(cache.embed_tokens(batches[0], wte=wte, wpe=wpe), ref))
        e = {"tokens": B * T, "announced_output_identical": same}
        for name in rows:
            e[name] = {"ms_per_step": float(np.median([r[0] for r in rows[name]])), "kernel_ms": float(np.median([r[1] for r in rows[name]])),
                       "all_ms_per_step": [round(r[0], 4) for r in rows[name]]}
        e["announced_over_serial"] = e["announced"]["ms_per_step"] / e["serial"]["ms_per_step"]
        res["sizes"][str(B)] = e
        sys.stderr.write(f"B={B}: serial {e['serial']['ms_per_step']:.4f} ms (kernel {e['serial']['kernel_ms']:.4f}), announced "
                         f"{e['announced']['ms_per_step']:.4f} ms (kernel {e['announced']['kernel_ms']:.4f}), ratio {e['announced_over_serial']:.3f}\n")
        del batches, out
    print(json.dumps(res))


if __name__ == "__main__":
    main()
