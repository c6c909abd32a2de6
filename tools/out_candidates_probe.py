import os, sys
sys.path.insert(0, "/root/repo")
import torch, bench
from scone_amd import EmbeddingCache
from scone_amd import synthetic as S
d, B, T = 768, 2048, 512
vocab_obj, keys, lens = bench.make_vocabulary(1_000_000, "zipf")
cache = EmbeddingCache.from_synthetic(vocab_obj, d, table_format="int8", seed=7, base_scale=0.02 / 127)
g = torch.Generator(device="cuda").manual_seed(5)
wte = (torch.randn(S.GPT2_VOCAB, d, generator=g, device="cuda") * 0.02).half()
wpe = (torch.randn(1024, d, generator=g, device="cuda") * 0.01).half()
_, batches = bench.make_batches(vocab_obj, keys, lens, "uniform", B, T, 1234, 4)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 16
out, rep = cache.alloc_output(batches[0], wte=wte, wpe=wpe, candidates=n, trials=3)
print("candidates in allocation order:", " ".join("%.4f" % t for t in rep["kernel_ms"]), "kept", rep["kept"], flush=True)
free, total = torch.cuda.mem_get_info()
print("free %.1f GB of %.1f" % (free / 1e9, total / 1e9))
