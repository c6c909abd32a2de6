#!/usr/bin/env python3
"""In-process A/B of the two large-batch lookup kernels: k_embed_wave (a wave = one position, walked over sequences; per-token
id records) against k_embed_slide (a wave = consecutive positions of one sequence, the bigram / trigram rows of its sliding
window in registers; per-position window hits), switched per call by SCONE_SLIDE.  Same table, same batches, alternating loops;
outputs compared bit for bit.

NOTE: k_embed_slide was removed after this measurement (3-10 % slower, no traffic saved: profiles/r05m, r05n); to repeat it
apply profiles/r05n/sliding_window_lookup_experiment.diff first -- on the current tree SCONE_SLIDE does nothing.

    python tools/slide_ab.py [--format int8 --dim 768 --rows 1000000 --keygen zipf] [--steps 30] [--rounds 3] [--seg 64]"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--format", default="int8")
    ap.add_argument("--dim", type=int, default=768)
    ap.add_argument("--rows", type=int, default=1_000_000)
    ap.add_argument("--keygen", default="zipf")
    ap.add_argument("--stream", default="uniform")
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--batch", type=int, default=2048)
    ap.add_argument("--seq", type=int, default=512)
    ap.add_argument("--segs", default="64")
    a = ap.parse_args()
    import numpy as np
    import torch
    import bench
    from scone_amd import EmbeddingCache
    from scone_amd import synthetic as S
    vocab_obj, keys, lens = bench.make_vocabulary(a.rows, a.keygen)
    kw = {"n_rows": a.rows} if keys is None else {}
    cache = EmbeddingCache.from_synthetic(vocab_obj, a.dim, table_format=a.format, seed=7, base_scale=0.02 / 127, **kw)
    g = torch.Generator(device="cuda").manual_seed(5)
    wte = (torch.randn(S.GPT2_VOCAB, a.dim, generator=g, device="cuda") * 0.02).half()
    wpe = (torch.randn(1024, a.dim, generator=g, device="cuda") * 0.01).half()
    B, T = a.batch, a.seq
    _, batches = bench.make_batches(vocab_obj, keys, lens, a.stream, B, T, 1234, min(a.steps + 4, 40))
    out = torch.empty(B, T, a.dim, dtype=torch.float16, device="cuda")
    os.environ["SCONE_SLIDE"] = "0"
    ref16 = cache.embed_tokens(batches[0], wte=wte, wpe=wpe).clone()
    ref32 = cache.embed_tokens(batches[0], out_dtype=torch.float32).clone()
    res = {"what": __doc__.split("\n\n")[0], "table": f"{a.rows}-row {a.format} d={a.dim} ({a.keygen})", "stream": a.stream, "tokens": B * T, "forms": {}}
    forms = [("wave", "0", None)] + [(f"slide_seg{s}", "1", s) for s in a.segs.split(",")]
    rows = {n: [] for n, _, _ in forms}
    same = {}
    for name, sw, seg in forms:
        os.environ["SCONE_SLIDE"] = sw
        if seg:
            os.environ["SCONE_SLIDE_SEG"] = str(seg)
        o16 = cache.embed_tokens(batches[0], wte=wte, wpe=wpe)
        o32 = cache.embed_tokens(batches[0], out_dtype=torch.float32)
        same[name] = {"fp16_bytes_equal": bool(torch.equal(o16, ref16)), "fp32_bits_equal": bool(torch.equal(o32, ref32)),
                      "status": int(cache.table.status())}
    for _ in range(a.rounds):
        for name, sw, seg in forms:
            os.environ["SCONE_SLIDE"] = sw
            if seg:
                os.environ["SCONE_SLIDE_SEG"] = str(seg)
            dt, nl, km, sm = bench.lookup_loop(cache, batches, wte, wpe, out, a.steps, 4, torch.cuda.synchronize, False)
            rows[name].append((dt / a.steps * 1e3, km / max(nl, 1)))
    for name in rows:
        res["forms"][name] = {"ms_per_step": float(np.median([r[0] for r in rows[name]])), "kernel_ms": float(np.median([r[1] for r in rows[name]])),
                              "all_ms_per_step": [round(r[0], 4) for r in rows[name]], **same[name]}
        sys.stderr.write(f"{name}: step {res['forms'][name]['ms_per_step']:.4f} ms, kernel {res['forms'][name]['kernel_ms']:.4f} ms, {same[name]}\n")
    os.environ["SCONE_SLIDE"] = "0"
    print(json.dumps(res))


if __name__ == "__main__":
    main()
