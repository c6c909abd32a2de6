import time, numpy as np, torch, sys
sys.path.insert(0, '.')
from scone_amd import NGramExtractor
from scone_amd import synthetic as S
rng = np.random.default_rng(1234)
cdf = S.zipf_cdf(50257)
corpus = [S.zipf_tokens(rng, cdf, 1000).tolist() for _ in range(1000)]       # 1M tokens (SURVEY 8d C1 corpus)
t0 = time.perf_counter(); host = NGramExtractor(max_n=3, min_freq=1, max_f_grams=100_000).fit(corpus, verbose=False); th = time.perf_counter() - t0
dev = NGramExtractor(max_n=3, min_freq=1, max_f_grams=100_000).fit_gpu(corpus, verbose=False)   # warm-up (allocations, first launch)
t0 = time.perf_counter(); dev = NGramExtractor(max_n=3, min_freq=1, max_f_grams=100_000).fit_gpu(corpus, verbose=False); td = time.perf_counter() - t0
hk, hl = host.key_arrays(); dk, dl = dev.key_arrays()
print("equal", np.array_equal(hk, dk) and np.array_equal(hl, dl), "host fit %.2f s  gpu fit (incl. list->array + H2D) %.3f s" % (th, td))
from scone_amd.hip_backend import fit_gpu
flat = torch.from_numpy(np.concatenate([np.asarray(c) for c in corpus])).cuda().int()
off = torch.arange(0, 1000 * 1000 + 1, 1000).cuda()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(5): fit_gpu(flat, off, 3, 1, 100_000)
torch.cuda.synchronize(); print("scone_fit device-resident corpus: %.1f ms per 1M tokens" % ((time.perf_counter() - t0) / 5 * 1e3))
