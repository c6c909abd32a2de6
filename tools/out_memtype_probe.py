#!/usr/bin/env python3
"""Round 6: does the MEMORY TYPE of the output buffer matter?  The lookup writes 1.6 GB per launch that nothing reads back; no
store flavour of gfx950 skips L2 allocation, but a PAGE attribute can: hipExtMallocWithFlags(hipDeviceMallocUncached /
hipDeviceMallocFinegrained).  The headline lookup through the bare ABI (scone_embed with a raw output pointer) into buffers of
each type, several of each (placement varies per allocation), HIP-event kernel time."""
import ctypes as C
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
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
hip = C.CDLL("libamdhip64.so")
hip.hipExtMallocWithFlags.argtypes = [C.POINTER(C.c_void_p), C.c_size_t, C.c_uint]
hip.hipFree.argtypes = [C.c_void_p]
nbytes = B * T * d * 2
FLAGS = {"default": 0x0, "finegrained": 0x1, "uncached": 0x3, "contiguous": 0x4}


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


ref = torch.empty(B, T, d, dtype=torch.float16, device="cuda")
cache.embed_tokens(batches[0], wte=wte, wpe=wpe, out=ref)
res = {}
for name, fl in FLAGS.items():
    times, ok = [], True
    ptrs = []
    for _ in range(4):
        p = C.c_void_p()
        rc = hip.hipExtMallocWithFlags(C.byref(p), nbytes, fl)
        if rc != 0 or not p.value:
            times.append(None)
            continue
        ptrs.append(p)
        times.append(run(p.value))
    if ptrs:                                               # the result written into the last buffer of this type is the lookup's
        stream = torch.cuda.current_stream().cuda_stream
        L.lib().scone_embed(table._h, batches[0].data_ptr(), B, T, wte.data_ptr(), wte.shape[0], wpe.data_ptr(), wpe.shape[0], None,
                            L.REDUCE_MEAN, ptrs[-1].value, L.DT_F16, stream)
        torch.cuda.synchronize()
        got = torch.empty_like(ref)
        hipMemcpy = hip.hipMemcpy
        hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
        hipMemcpy(got.data_ptr(), ptrs[-1].value, nbytes, 3)   # device to device
        torch.cuda.synchronize()
        ok = bool(torch.equal(got, ref))
    for p in ptrs:
        hip.hipFree(p)
    res[name] = {"kernel_ms": times, "result_equal": ok}
    print(name, times, ok, flush=True)
torch_bufs = [torch.empty(B, T, d, dtype=torch.float16, device="cuda") for _ in range(4)]
res["torch.empty"] = {"kernel_ms": [run(t.data_ptr()) for t in torch_bufs]}
print("torch.empty", res["torch.empty"]["kernel_ms"], flush=True)
print(json.dumps(res))
