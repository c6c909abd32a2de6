import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from scone_amd import EmbeddingCache, NGramExtractor
from scone_amd import synthetic as S
d = 768
keys, lens = S.make_keys(100_000, S.GPT2_VOCAB, 3, seed=11)
ex = NGramExtractor.from_arrays(keys, lens, max_n=3)
cache = EmbeddingCache.from_synthetic(ex, d, table_format="fp32")
seq = S.stream_zipf(S.GPT2_VOCAB, 1, 512, 3)[0].tolist()
for name, fn in (("get_token_embeddings(512 tokens) -> {pos: Tensor[K,d]} on CPU", lambda: cache.get_token_embeddings(seq)),
                 ("get_token_embeddings(512 tokens, device=cuda)", lambda: cache.get_token_embeddings(seq, device=torch.device("cuda"))),
                 ("NGramExtractor.get_token_f_grams(512 tokens)", lambda: ex.get_token_f_grams(seq)),
                 ("get_embeddings(6 ids)", lambda: cache.get_embeddings([1, 2, 3, 4, 5, 6]))):
    fn(); fn()
    t0 = time.perf_counter(); n = 0
    while time.perf_counter() - t0 < 1.0:
        fn(); n += 1
    print(f"{name}: {(time.perf_counter() - t0) / n * 1e3:.3f} ms per call", flush=True)
