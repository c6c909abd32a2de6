#!/usr/bin/env python3
"""What do the library's timing events (scone_profile_enable: two hipEventRecord per lookup) cost the step?  The headline
loop with and without them, alternating in one process.   python tools/prof_event_cost.py [--steps 40] [--rounds 4]"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--rounds", type=int, default=4)
    a = ap.parse_args()
    import numpy as np
    import torch
    import bench
    from scone_amd import EmbeddingCache
    from scone_amd import synthetic as S
    vocab_obj, keys, lens = bench.make_vocabulary(1_000_000, "zipf")
    cache = EmbeddingCache.from_synthetic(vocab_obj, 768, table_format="int8", seed=7, base_scale=0.02 / 127)
    g = torch.Generator(device="cuda").manual_seed(5)
    wte = (torch.randn(S.GPT2_VOCAB, 768, generator=g, device="cuda") * 0.02).half()
    wpe = (torch.randn(1024, 768, generator=g, device="cuda") * 0.01).half()
    B, T = 2048, 512
    _, batches = bench.make_batches(vocab_obj, keys, lens, "uniform", B, T, 1234, a.steps + 5)
    out = torch.empty(B, T, 768, dtype=torch.float16, device="cuda")
    cache.table.reserve(B * T)
    rows = {"events_on": [], "events_off": []}

    def loop(n):
        for i in range(n):
            cache.embed_tokens(batches[i % len(batches)], wte=wte, wpe=wpe, out=out)
    for _ in range(a.rounds):
        for name, on in (("events_off", False), ("events_on", True)):
            cache.table.profile_enable(on)
            loop(5)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            loop(a.steps)
            torch.cuda.synchronize()
            rows[name].append((time.perf_counter() - t0) / a.steps * 1e3)
            cache.table.profile_read(reset=True)
            cache.table.profile_enable(False)
    print(json.dumps({"what": __doc__.split("\n\n")[0], "steps": a.steps,
                      "ms_per_step": {k: {"median": float(np.median(v)), "all": [round(x, 4) for x in v]} for k, v in rows.items()},
                      "events_cost_us": (float(np.median(rows["events_on"])) - float(np.median(rows["events_off"]))) * 1e3}))


if __name__ == "__main__":
    main()
