#!/usr/bin/env python3
"""Round 6: is a deliberately SCATTERED output buffer deterministically fast?  tools/probes/libscatter_alloc.so builds a buffer out
of physical chunks mapped into one virtual range in creation order / shuffled / reversed order (HIP virtual memory management);
the headline lookup writes into it through the bare ABI.  Controls: plain torch allocations."""
import ctypes as C
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from scone_amd import EmbeddingCache, _lib as L
from scone_amd import synthetic as S

d, B, T = 768, 2048, 512
vocab_obj, keys, lens = bench.make_vocabulary(1_000_000, "zipf")
cache = EmbeddingCache.from_synthetic(vocab_obj, d, table_format="int8", seed=7, base_scale=0.02 / 127)
g = torch.Generator(device="cuda").manual_seed(5)
wte = (torch.randn(S.GPT2_VOCAB, d, generator=g, device="cuda") * 0.02).half()
wpe = (torch.randn(1024, d, generator=g, device="cuda") * 0.01).half()
_, batches = bench.make_batches(vocab_obj, keys, lens, "uniform", B, T, 1234, 6)
table = cache.table
sa = C.CDLL(os.path.join(ROOT, "tools", "probes", "libscatter_alloc.so"))
sa.scatter_alloc.restype = C.c_void_p
sa.scatter_alloc.argtypes = [C.c_size_t, C.c_size_t, C.c_int, C.c_uint, C.c_int]
sa.scatter_free.argtypes = [C.c_void_p]
sa.scatter_granularity.restype = C.c_size_t
nbytes = B * T * d * 2
print("allocation granularity", sa.scatter_granularity(0), flush=True)


def run(ptr, n=12):
    stream = torch.cuda.current_stream().cuda_stream
    def call(tok):
        rc = L.lib().scone_embed(table._h, tok.data_ptr(), B, T, wte.data_ptr(), wte.shape[0], wpe.data_ptr(), wpe.shape[0], None,
                                 L.REDUCE_MEAN, ptr, L.DT_F16, stream)
        assert rc == 0, L.lib().scone_last_error(table._h)
    for i in range(3):
        call(batches[i % 6])
    table.profile_enable(True); table.profile_read(reset=True)
    for i in range(n):
        call(batches[i % 6])
    k, ms = table.profile_read(reset=True)
    table.profile_enable(False)
    return ms / k


res = {}
plain = [torch.empty(B, T, d, dtype=torch.float16, device="cuda") for _ in range(6)]
res["torch.empty x6"] = [run(t.data_ptr()) for t in plain]
print("torch.empty x6", ["%.4f" % x for x in res["torch.empty x6"]], flush=True)
for chunk_mb in (2, 8, 64):
    for mode, name in ((0, "in order"), (1, "shuffled"), (2, "reversed")):
        times = []
        for rep in range(3):
            p = sa.scatter_alloc(nbytes, chunk_mb << 20, mode, 1234 + rep, 0)
            if not p:
                times.append(None)
                continue
            times.append(run(p))
            torch.cuda.synchronize()
            sa.scatter_free(p)
        res[f"{chunk_mb} MB chunks, {name}"] = times
        print(f"{chunk_mb:3d} MB chunks, {name:9s}", ["%.4f" % x if x else None for x in times], flush=True)
res["torch.empty x6 again"] = [run(t.data_ptr()) for t in plain]
print("torch.empty again", ["%.4f" % x for x in res["torch.empty x6 again"]], flush=True)
print(json.dumps(res))
