#!/usr/bin/env python3
"""Does host -> HBM traffic moved by the COPY ENGINES slow the lookup kernel the way the cache's copy KERNEL does?
C4's table entirely in HBM, Zipf stream, a different batch every step; on a side stream `--mb-per-step` of pinned host memory
are copied to the device with hipMemcpyAsync (torch's non_blocking copy_ of a pinned tensor: the SDMA path) in `--pieces` pieces
per step, enqueued step by step beside the lookups.  Prints ms/step alone and with the copies in flight, alternating."""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from scone_amd import EmbeddingCache
from scone_amd import synthetic as S


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=100_000_000)
    ap.add_argument("--mb-per-step", type=float, default=24.0)
    ap.add_argument("--pieces", type=int, default=4)
    ap.add_argument("--kind", default="h2d", choices=["h2d", "d2d"], help="d2d: HBM -> HBM on this GPU by the copy engines "
                    "(hipMemcpyAsync, hipMemcpyDeviceToDeviceNoCU, through scone_ipc_push) on --streams streams: what the sdma "
                    "transport of the sharded step does to the memory system, minus the links")
    ap.add_argument("--streams", type=int, default=14)
    ap.add_argument("--src-mb", type=float, default=0.0, help="d2d: every piece reads from the SAME region of this many MB (it "
                    "stays in the Infinity Cache): the HBM traffic is then the writes alone -- what a rank's HBM sees of an "
                    "exchange whose incoming bytes are written by the peers' engines and whose 7 outgoing pushes re-read one buffer")
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--rounds", type=int, default=3)
    a = ap.parse_args()
    N, d, B, T = a.rows, 1024, 2048, 512
    vocab = S.StructuredVocab(N)
    cache = EmbeddingCache.from_synthetic(vocab, d, table_format="int4", seed=7, base_scale=0.02 / 127, n_rows=N)
    g = torch.Generator(device="cuda").manual_seed(5)
    wte = (torch.randn(S.GPT2_VOCAB, d, generator=g, device="cuda") * 0.02).half()
    wpe = (torch.randn(1024, d, generator=g, device="cuda") * 0.01).half()
    out = torch.empty(B, T, d, dtype=torch.float16, device="cuda")
    toks = [S.stream_zipf_ids_torch(vocab, B, T, 1234 + i) for i in range(a.steps)]
    piece = int(a.mb_per_step * 1e6 / a.pieces)
    src = torch.empty(piece * a.pieces, dtype=torch.uint8).pin_memory()
    dst = torch.empty(piece * a.pieces, dtype=torch.uint8, device="cuda")
    side = torch.cuda.Stream()
    sides = [torch.cuda.Stream() for _ in range(a.streams)]
    dsrc = torch.empty(piece * a.pieces, dtype=torch.uint8, device="cuda") if a.kind == "d2d" else None
    src_span = max(int(a.src_mb * 1e6), piece) if a.src_mb else piece * a.pieces
    cache.table.reserve(B * T)
    for t in toks[:3]:
        cache.embed_tokens(t, wte=wte, wpe=wpe, out=out)
    torch.cuda.synchronize()

    res = {"alone_ms": [], "with_copies_ms": []}

    def run(copies):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(a.steps):
            cache.embed_tokens(toks[i], wte=wte, wpe=wpe, out=out)
            if copies and a.kind == "d2d":                    # free-running: the engines are busy for the whole loop
                for k in range(a.pieces):
                    with torch.cuda.stream(sides[k % len(sides)]):
                        so = (k * piece) % max(src_span - piece + 1, 1) if a.src_mb else k * piece
                        cache.table.ipc_push(dst.data_ptr() + k * piece, dsrc.data_ptr() + so, piece, True)
            elif copies:
                ev = torch.cuda.Event()
                ev.record()                                   # the copies of step i start when lookup i - 1 is done: beside lookup i
                side.wait_event(ev)
                with torch.cuda.stream(side):
                    for k in range(a.pieces):
                        dst[k * piece:(k + 1) * piece].copy_(src[k * piece:(k + 1) * piece], non_blocking=True)
        cur = torch.cuda.current_stream()
        cur.synchronize()                                     # the LOOKUPS' time; the copies may still be draining
        dt = (time.perf_counter() - t0) / a.steps * 1e3
        torch.cuda.synchronize()
        res.setdefault("copies_drained_after_ms", []).append((time.perf_counter() - t0) * 1e3 if copies else 0.0)
        return dt

    for _ in range(a.rounds):
        res["alone_ms"].append(run(False))
        res["with_copies_ms"].append(run(True))
    res.update(kind=a.kind, mb_per_step=a.mb_per_step, pieces=a.pieces, steps=a.steps, rows=N, streams=a.streams, src_mb=a.src_mb)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
