#!/usr/bin/env python3
"""Round 6: is the output buffer's placement effect (profiles/r06m) a TLB effect?  Six separate 1.6-GB output buffers, per buffer
2 warm-up + 4 timed lookups of the headline workload (HIP-event kernel time printed per buffer).  Run plain, and under
`rocprofv3 --pmc TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_REQUEST_sum --kernel-trace`: the k_embed_wave dispatches come buffer by
buffer, 6 per buffer (tools/runs/r06n.sh splits the counter file by that order)."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from scone_amd import EmbeddingCache, NGramExtractor
from scone_amd import synthetic as S

d, N, B, T = 768, 1_000_000, 2048, 512
keys, lens = S.make_keys(N, S.GPT2_VOCAB, 3, seed=11)
ex = NGramExtractor.from_arrays(keys, lens, max_n=3)
toks = [torch.from_numpy(S.stream_uniform_ids(keys, lens, B, T, 1234 + 7919 * i)).to("cuda", torch.int32) for i in range(6)]
g = torch.Generator(device="cuda").manual_seed(5)
wte = (torch.randn(S.GPT2_VOCAB, d, generator=g, device="cuda") * 0.02).half()
wpe = (torch.randn(1024, d, generator=g, device="cuda") * 0.01).half()
cache = EmbeddingCache.from_synthetic(ex, d, table_format="int8")
outs = [torch.empty(B, T, d, dtype=torch.float16, device="cuda") for _ in range(6)]
res = []
for o in outs:
    for i in range(2):
        cache.embed_tokens(toks[i], wte=wte, wpe=wpe, out=o)
    cache.table.profile_enable(True); cache.table.profile_read(reset=True)
    for i in range(4):
        cache.embed_tokens(toks[2 + i], wte=wte, wpe=wpe, out=o)
    k, ms = cache.table.profile_read(reset=True)
    cache.table.profile_enable(False)
    res.append(ms / k)
print(json.dumps({"kernel_ms_per_buffer": res, "dispatches_per_buffer": 6}))
