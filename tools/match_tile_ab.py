#!/usr/bin/env python3
"""k_match_ell against the positions-per-workgroup it is launched with (SCONE_MATCH_TILE; default: whole residency
rounds), headline workload, alternating in one process: stream time of `scone_embed` steps minus their gather-kernel time
(= match + launch gap), and the whole step."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from scone_amd import EmbeddingCache, NGramExtractor
from scone_amd import synthetic as S


def main():
    keys, lens = S.make_keys(1_000_000, S.GPT2_VOCAB, 3, seed=11)
    ex = NGramExtractor.from_arrays(keys, lens, max_n=3)
    B, T, d = 2048, 512, 768
    tok = torch.from_numpy(S.stream_uniform_ids(keys, lens, B, T, 1234)).to("cuda", torch.int32)
    g = torch.Generator(device="cuda").manual_seed(5)
    wte = (torch.randn(S.GPT2_VOCAB, d, generator=g, device="cuda") * 0.02).half()
    wpe = (torch.randn(1024, d, generator=g, device="cuda") * 0.01).half()
    out = torch.empty(B, T, d, dtype=torch.float16, device="cuda")
    caches = {}
    for tile in sys.argv[1:] or ["0", "254", "171", "128"]:
        os.environ["SCONE_MATCH_TILE"] = tile                       # read when the handle is created
        caches[tile] = EmbeddingCache.from_synthetic(ex, d, table_format="int8", seed=7, base_scale=0.02 / 127)
    res = {t: [] for t in caches}
    ref = None
    for rnd in range(4):
        for tile, c in caches.items():
            table = c.table
            for _ in range(5):
                c.embed_tokens(tok, wte=wte, wpe=wpe, out=out)
            table.profile_enable(True)
            table.profile_read(reset=True)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(50):
                c.embed_tokens(tok, wte=wte, wpe=wpe, out=out)
            e1.record()
            torch.cuda.synchronize()
            n, km = table.profile_read(reset=True)
            table.profile_enable(False)
            res[tile].append({"step_us": e0.elapsed_time(e1) / 50 * 1e3, "match_plus_gap_us": (e0.elapsed_time(e1) - km) / 50 * 1e3})
            chk = out.float().abs().sum().item()
            ref = chk if ref is None else ref
            assert chk == ref
    print(json.dumps({t: {"step_us": sorted(x["step_us"] for x in v), "match_plus_gap_us": sorted(x["match_plus_gap_us"] for x in v)}
                      for t, v in res.items()}))


if __name__ == "__main__":
    main()
