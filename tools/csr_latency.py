import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from scone_amd import EmbeddingCache, NGramExtractor
from scone_amd import synthetic as S
d, N, B, T = 768, 1_000_000, 2048, 512
keys, lens = S.make_keys(N, S.GPT2_VOCAB, 3, seed=11)
cache = EmbeddingCache.from_synthetic(NGramExtractor.from_arrays(keys, lens, max_n=3), d, table_format="int8")
table = cache.table
tok = torch.from_numpy(S.stream_uniform_ids(keys, lens, B, T, 1234)).to("cuda", torch.int32)
off, ids = table.match_csr(tok)
base = torch.randn(B * T, d, device="cuda").half()
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
print("scone_match_csr           %.3f ms" % timeit(lambda: table.match_csr(tok)))
print("scone_gather_reduce (CSR) %.3f ms (fp16 out + base)" % timeit(lambda: table.gather_reduce(off, ids, "mean", base=base, out_dtype=torch.float16)))
print("scone_gather_reduce (CSR) %.3f ms (fp32 out, no base)" % timeit(lambda: table.gather_reduce(off, ids, "mean")))
wte = (torch.randn(S.GPT2_VOCAB, d, device="cuda") * 0.02).half(); wpe = (torch.randn(1024, d, device="cuda") * 0.01).half()
out = torch.empty(B, T, d, dtype=torch.float16, device="cuda")
print("scone_embed (fused path)  %.3f ms" % timeit(lambda: cache.embed_tokens(tok, wte=wte, wpe=wpe, out=out)))
