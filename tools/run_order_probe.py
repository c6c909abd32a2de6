#!/usr/bin/env python3
"""Traffic probe of a TOKEN-ORDERED ("run-ordered") lookup, measured before any such kernel is written (round 6).

Needs a -DSCONE_PROBE_PERM build of the library (tools/build_variant.sh perm -DSCONE_PROBE_PERM; SCONE_HIP_LIB points at it):
that build's scone_embed looks up, at slot j of the launch, position perm[j] (record, token, position row) and stores to
out[j] -- see k_probe_permute in scone_gather.hip.  This script lays the permutation out so that the positions ONE WAVE of
k_embed_wave visits one after the other are a run of positions with the same token id (runs cut at the wave's trip count,
43-49 positions), i.e. the visiting order of the review's "counting sort by token id, a wave walks a run".  The within-run
re-reads of the token's wte / unigram rows then hit L1 / L2 instead of living in registers, so FETCH_SIZE (bytes past L2) is
what a register-caching kernel would fetch, and the kernel time an upper bound on its time.

Dispatch order of k_embed_wave (tools/runs/r06a.sh splits the counter files by it):
    warm-up x 3, then N x [normal], N x [positions given, position order], N x [positions given, token order]
  normal          the shipped form: position row parked in LDS, 8 waves / SIMD
  position order  the same visiting order, position ids passed explicitly (the per-token wpe kernel, 7 waves / SIMD): the control
  token order     the permutation above

    SCONE_HIP_LIB=gpurun_ab/libperm.so python tools/run_order_probe.py [--steps 6] [--stream uniform|zipf]"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def wave_walk_layout(B, T, cus, waves, rounds=3):
    """(seqs_per_block, chunks) of launch_wave (scone_embed_wave.h) for the per-token-position kernel."""
    pos_groups = (T + 3) // 4
    target = rounds * cus * waves
    chunks = max(1, min(B, (target + pos_groups // 2) // pos_groups))
    spb = (B + chunks - 1) // chunks
    chunks = (B + spb - 1) // spb
    return spb, chunks


def token_order_perm(tok, B, T, spb, chunks):
    """perm[j] = the position looked up at slot j, such that wave (chunk c, position slot i), which visits the slots
    j = (b0 + k) * T + i for k = 0, 1, ..., walks CONSECUTIVE entries of the token-sorted position list."""
    import torch
    flat = tok.reshape(-1).to(torch.int64)
    order = torch.sort(flat * (B * T) + torch.arange(B * T, device=tok.device), stable=True).indices  # by (token, position)
    perm = torch.empty(B * T, dtype=torch.int32, device=tok.device)
    for c in range(chunks):
        b0, b1 = c * spb, min(B, (c + 1) * spb)
        n = b1 - b0
        seg = order[b0 * T:b1 * T].view(T, n)          # [i, k]: wave i of the chunk takes n consecutive sorted positions
        perm.view(B, T)[b0:b1] = seg.t().to(torch.int32)  # slot (b0 + k, i)
    return perm


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=6)
    ap.add_argument("--stream", default="uniform")
    a = ap.parse_args()
    import torch
    import bench
    from scone_amd import EmbeddingCache
    from scone_amd import synthetic as S
    d, B, T = 768, 2048, 512
    vocab_obj, keys, lens = bench.make_vocabulary(1_000_000, "zipf")
    cache = EmbeddingCache.from_synthetic(vocab_obj, d, table_format="int8", seed=7, base_scale=0.02 / 127)
    g = torch.Generator(device="cuda").manual_seed(5)
    wte = (torch.randn(S.GPT2_VOCAB, d, generator=g, device="cuda") * 0.02).half()
    wpe = (torch.randn(1024, d, generator=g, device="cuda") * 0.01).half()
    _, batches = bench.make_batches(vocab_obj, keys, lens, a.stream, B, T, 1234, a.steps + 1)
    out = torch.empty(B, T, d, dtype=torch.float16, device="cuda")
    cus = torch.cuda.get_device_properties(0).multi_processor_count
    spb, chunks = wave_walk_layout(B, T, cus, waves=7)
    pos = torch.arange(T, dtype=torch.int32, device="cuda").repeat(B, 1).contiguous()
    perms = [token_order_perm(b, B, T, spb, chunks) for b in batches]
    runs = []
    for b, pm in zip(batches[:2], perms[:2]):       # run statistics of the visiting order
        tv = b.reshape(-1)[pm.long()].view(B, T)
        walk = torch.cat([tv[c * spb:min(B, (c + 1) * spb)].t().reshape(-1) for c in range(chunks)])
        changes = int((walk[1:] != walk[:-1]).sum()) + 1
        runs.append({"token_changes_along_the_walks": changes, "distinct_tokens": int(torch.unique(b).numel())})
    # the permuted launch is the right vectors in permuted places: check one batch against the normal launch
    os.environ.pop("SCONE_PROBE_PERM_PTR", None)
    ref = torch.empty_like(out)
    cache.embed_tokens(batches[0], wte=wte, wpe=wpe, out=ref)
    os.environ["SCONE_PROBE_PERM_PTR"] = hex(perms[0].data_ptr())
    cache.embed_tokens(batches[0], wte=wte, wpe=wpe, out=out)
    torch.cuda.synchronize()
    same = bool(torch.equal(out.view(B * T, d), ref.view(B * T, d)[perms[0].long()]))
    os.environ.pop("SCONE_PROBE_PERM_PTR", None)
    cache.embed_tokens(batches[-1], wte=wte, wpe=wpe, out=out)     # third warm-up launch
    torch.cuda.synchronize()
    times = {}
    for name in ("normal", "position_order", "token_order"):
        cache.table.profile_enable(True)
        cache.table.profile_read(reset=True)
        for i in range(a.steps):
            os.environ.pop("SCONE_PROBE_PERM_PTR", None)
            kw = {}
            if name == "position_order":
                kw["position_ids"] = pos
            elif name == "token_order":
                os.environ["SCONE_PROBE_PERM_PTR"] = hex(perms[i].data_ptr())
            cache.embed_tokens(batches[i], wte=wte, wpe=wpe, out=out, **kw)
        torch.cuda.synchronize()
        v = cache.table.profile_samples()
        times[name] = {"kernel_ms_median": float(sorted(v)[len(v) // 2]), "kernel_ms_min": float(min(v)), "launches": int(len(v))}
        cache.table.profile_enable(False)
    os.environ.pop("SCONE_PROBE_PERM_PTR", None)
    print(json.dumps({"steps": a.steps, "stream": a.stream, "seqs_per_block": spb, "chunks": chunks, "cus": cus,
                      "permuted_launch_equals_permuted_reference": same, "visiting_order": runs, "kernel_times": times,
                      "order": ["warm x 3", f"{a.steps} x normal", f"{a.steps} x position_order", f"{a.steps} x token_order"]}))


if __name__ == "__main__":
    main()
