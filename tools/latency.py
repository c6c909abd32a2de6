import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from scone_amd import EmbeddingCache, NGramExtractor
from scone_amd import synthetic as S
d = 768
keys, lens = S.make_keys(1_000_000, S.GPT2_VOCAB, 3, seed=11)
cache = EmbeddingCache.from_synthetic(NGramExtractor.from_arrays(keys, lens, max_n=3), d, table_format="int8")
wte = (torch.randn(S.GPT2_VOCAB, d, device="cuda") * 0.02).half(); wpe = (torch.randn(1024, d, device="cuda") * 0.01).half()
for B, T in [tuple(int(x) for x in s.split("x")) for s in os.environ.get("SHAPES", "1x512,4x512,8x512,4x1024,8x1024,16x1024").split(",")]:
    tok = torch.from_numpy(S.stream_zipf(S.GPT2_VOCAB, B, T, 99)).to("cuda", torch.int32)
    out = torch.empty(B, T, d, dtype=torch.float16, device="cuda")
    for _ in range(50): cache.embed_tokens(tok, wte=wte, wpe=wpe, out=out)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    n = 2000
    for _ in range(n): cache.embed_tokens(tok, wte=wte, wpe=wpe, out=out)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
    print(f"{os.path.basename(os.environ.get('SCONE_HIP_LIB', 'default'))}: {B} x {T}: {dt * 1e6:.1f} us per call, {B * T / dt / 1e6:.0f} M tok/s", flush=True)
