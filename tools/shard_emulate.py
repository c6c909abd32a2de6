#!/usr/bin/env python3
"""The W-way row exchange at scale on ONE GPU: W real shards of a table that fits one GPU only in total
(default: 100M INT4 rows d = 1024 = 52.8 GB over 8 handles, each with its own replicated index), a 1M-token
batch, the all-to-all done by hand (device copies, not timed).  Measures what one GPU can measure of the
8-GPU step: every rank's LOCAL phases (plan = 2 matches + counts, pack, embed of its slice) with HIP events,
the record counts per (source, destination) pair and the wire bytes; checks the assembled output against
the unsharded lookup of the same tokens when the table also fits as ONE handle (--check).

    python tools/shard_emulate.py [--rows 100000000] [--world 8] [--check]

The xGMI time itself cannot be measured here; DESIGN.md prices it from the printed byte counts.
"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from scone_amd import synthetic as S
from scone_amd.distributed import shard_range
from scone_amd.hip_backend import SconeTable


def timed(fn):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    a.record()
    out = fn()
    b.record()
    torch.cuda.synchronize()
    return out, a.elapsed_time(b)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=100_000_000)
    ap.add_argument("--world", type=int, default=8)
    ap.add_argument("--dim", type=int, default=1024)
    ap.add_argument("--format", default="int4")
    ap.add_argument("--batch", type=int, default=2048)
    ap.add_argument("--seq", type=int, default=512)
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--check", action="store_true")
    ap.add_argument("--head", type=int, default=0, help="replicated head: global rows [0, head) kept on every shard")
    ap.add_argument("--chunks", type=int, default=1, help="gather_rows: chunks of the pipelined exchange (1 = one shot)")
    ap.add_argument("--side-high-priority", action="store_true", help="--wall: the split-phase loop's side stream gets high priority")
    ap.add_argument("--wall", action="store_true",
                    help="gather_rows: also time rank 0's whole step back to back (no phase synchronisation): the local "
                         "critical path including launch gaps and the plan's one host synchronisation")
    ap.add_argument("--mode", default="rows_dedup", choices=["rows_dedup", "gather_rows", "gather_cols"],
                    help="rows_dedup: all-to-all of records (one per distinct row and destination), every rank reduces its slice; gather_rows: all-gather of records, "
                         "every rank reduces the whole batch; gather_cols: the same with columns on the wire (payload rows | "
                         "scales | the senders' hash fragments) and the plan's match sharded over the ranks")
    a = ap.parse_args()
    N, W, d, B, T = a.rows, a.world, a.dim, a.batch, a.seq
    keys, lens = S.make_keys_structured(N, S.GPT2_VOCAB, 3) if N >= 20_000_000 else S.make_keys(N, S.GPT2_VOCAB, 3, seed=11)
    tok_np = S.stream_uniform_ids(keys, lens, B, T, 1234)
    tok = torch.from_numpy(tok_np).to("cuda", torch.int32)
    g = torch.Generator(device="cuda").manual_seed(5)
    wte = (torch.randn(S.GPT2_VOCAB, d, generator=g, device="cuda") * 0.02).half()
    wpe = (torch.randn(1024, d, generator=g, device="cuda") * 0.01).half()
    shards = []
    for r in range(W):
        lo, hi = shard_range(N, r, W)
        s = SconeTable(3, N, d, a.format, row_begin=lo, row_end=hi)
        s.index_build(keys, lens)
        if a.head:
            s.shard_set_head(a.head)
        s.fill_synthetic(7, 0.02 / 127)                                  # fills the head as well
        shards.append(s)
    rec = shards[0].shard_record_bytes()
    bper = (B + W - 1) // W
    res = {"rows": N, "world": W, "format": a.format, "dim": d, "tokens": B * T, "record_bytes": rec,
           "replicated_head_rows": a.head, "ranks": []}
    out = torch.empty(B * T, d, dtype=torch.float16, device="cuda")
    best = None
    if a.mode == "rows_dedup":
        rows_dedup_mode(a, shards, tok, wte, wpe, out, res, keys, lens)
        return
    if a.mode == "gather_rows" and (a.chunks > 1 or a.wall):
        gather_rows_chunked(a, shards, tok, wte, wpe, out, res, keys, lens)
        return
    if a.mode == "gather_rows":
        gather_rows_mode(a, shards, tok, wte, wpe, out, res, keys, lens)
        return
    if a.mode == "gather_cols":
        gather_cols_mode(a, shards, tok, wte, wpe, out, res, keys, lens)
        return


def gather_rows_mode(a, shards, tok, wte, wpe, out, res, keys, lens):
    """Every rank packs one record per DISTINCT row it owns that the batch references, the records are all-gathered (here:
    concatenated), every rank indexes them by row id and reduces the whole batch."""
    N, W, d, B, T = a.rows, a.world, a.dim, a.batch, a.seq
    rec = res["record_bytes"]
    best = None
    for rep in range(a.reps):
        sends, t_plan, t_pack, t_embed = [], [], [], []
        for r, s in enumerate(shards):
            n_rec, ms = timed(lambda: s.shard_gather_plan(tok))
            t_plan.append(ms)
            buf, ms = timed(lambda: s.shard_gather_pack(n_rec))
            sends.append(buf)
            t_pack.append(ms)
        recv = torch.cat(sends).contiguous()                              # the all-gather of records, by hand
        for q in range(W):
            _, ms = timed(lambda: shards[q].shard_gather_embed(tok, recv, wte=wte, wpe=wpe, out_dtype=torch.float16, out=out))
            t_embed.append(ms)
        cur = [t_plan, t_pack, t_embed]
        best = cur if best is None else [[min(x, y) for x, y in zip(b, c)] for b, c in zip(best, cur)]
        counts = [int(x.shape[0]) for x in sends]
        del sends
    for r in range(W):
        res["ranks"].append({"rank": r, "plan_ms": best[0][r], "pack_ms": best[1][r], "embed_ms": best[2][r],
                             "records_contributed": counts[r]})
    loc = [x["plan_ms"] + x["pack_ms"] + x["embed_ms"] for x in res["ranks"]]
    res["mode"] = "gather_rows"
    res["local_ms_max"], res["local_ms_mean"] = max(loc), sum(loc) / len(loc)
    res["all_gather_bytes_into_each_rank"] = int((sum(counts) - min(counts)) * rec)
    res["all_gather_padded_bytes_total"] = int(max(counts) * rec * W)
    if a.check:
        full = SconeTable(3, N, d, a.format)
        full.index_build(keys, lens)
        full.fill_synthetic(7, 0.02 / 127)
        want = full.embed(tok, wte=wte, wpe=wpe).reshape(B * T, d)
        res["bit_identical_to_unsharded"] = bool(torch.equal(out, want))
    print(json.dumps(res))


def gather_cols_mode(a, shards, tok, wte, wpe, out, res, keys, lens):
    """Round 3's one-piece all-gather form: rank r matches slice r of the batch, the 32-B list records are all-gathered (here:
    written side by side), every rank claims its distinct rows from the gathered lists and packs them as COLUMNS -- payload
    rows, scales, and its own hash fragment row id -> position -- into its range of the three receive buffers; every rank then
    resolves the lists through the owners' fragments and reduces the whole batch.  No receiver indexes anything."""
    N, W, d, B, T = a.rows, a.world, a.dim, a.batch, a.seq
    s0 = shards[0]
    wd, pb, sb, head = s0.ell_width(), s0.payload_bytes(), s0.scale_bytes(), a.head
    bper = (B + W - 1) // W
    best = None
    for rep in range(a.reps):
        t_match, t_plan, t_pack, t_embed = [], [], [], []
        ell = torch.empty((W * bper * T, wd), dtype=torch.int32, device="cuda")
        for r, s in enumerate(shards):
            b0, b1 = min(r * bper, B), min(r * bper + bper, B)
            _, ms = timed(lambda: s.shard_gather_match(tok, b0, b1, ell[b0 * T:max(b1, b0) * T]))
            t_match.append(ms)
        counts, ells = [], []
        for s in shards:
            mine = ell.clone()                                                       # (every rank owns its gathered copy)
            n, ms = timed(lambda: s.shard_gather_plan_ell(mine, B, T, 1)[0])
            counts.append(n)
            ells.append(mine)
            t_plan.append(ms)
        slots = [SconeTable.cols_frag_slots(c) for c in counts]
        rb = [sum(counts[:r]) for r in range(W)]
        fo = [sum(slots[:r]) for r in range(W)]
        total = sum(counts)
        rows = torch.empty((max(total, 1), pb), dtype=torch.uint8, device="cuda")
        scales = torch.empty((head + max(total, 1), sb), dtype=torch.uint8, device="cuda") if sb else None
        frags = torch.empty(sum(slots), dtype=torch.int64, device="cuda")
        if scales is not None and head:
            s0.shard_head_scales_into(scales)
        for r, s in enumerate(shards):
            _, ms = timed(lambda: s.shard_cols_pack(0, counts[r], rows[rb[r]:rb[r] + counts[r]],
                                                    None if scales is None else scales[head + rb[r]:head + rb[r] + counts[r]],
                                                    frags[fo[r]:fo[r] + slots[r]]))
            t_pack.append(ms)
        for q, s in enumerate(shards):
            _, ms = timed(lambda: s.shard_cols_embed(tok, 0, B, rows, total, scales, frags, fo, slots, rb, out, wte=wte, wpe=wpe))
            t_embed.append(ms)
        cur = [t_match, t_plan, t_pack, t_embed]
        best = cur if best is None else [[min(x, y) for x, y in zip(b, c)] for b, c in zip(best, cur)]
        del ells
    for r in range(W):
        res["ranks"].append({"rank": r, "match_slice_ms": best[0][r], "claim_ms": best[1][r], "pack_ms": best[2][r],
                             "remap_and_embed_ms": best[3][r], "rows_contributed": counts[r], "fragment_slots": slots[r]})
    loc = [sum(best[k][r] for k in range(4)) for r in range(W)]
    res["mode"] = "gather_cols"
    res["local_ms_max"], res["local_ms_mean"] = max(loc), sum(loc) / len(loc)
    res["bytes_into_each_rank"] = {"rows": int((total - min(counts)) * pb), "scales": int((total - min(counts)) * sb),
                                   "fragments": int((sum(slots) - min(slots)) * 8), "list_records": int((W - 1) * bper * T * wd * 4)}
    if a.check:
        full = SconeTable(3, N, d, a.format)
        full.index_build(keys, lens)
        full.fill_synthetic(7, 0.02 / 127)
        want = full.embed(tok, wte=wte, wpe=wpe).reshape(B * T, d)
        res["bit_identical_to_unsharded"] = bool(torch.equal(out, want))
    print(json.dumps(res))


def rows_dedup_mode(a, shards, tok, wte, wpe, out, res, keys, lens):
    """The slice exchange with one record per distinct row and destination (ShardedEmbeddingCache._embed_row_exchange_dedup),
    the all-to-all done by hand."""
    N, W, d, B, T = a.rows, a.world, a.dim, a.batch, a.seq
    rec = res["record_bytes"]
    bper = (B + W - 1) // W
    best = None
    for rep in range(a.reps):
        ends, t_plan, sends, t_pack, t_embed = [], [], [], [], []
        for s in shards:
            e, ms = timed(lambda: s.shard_gather_plan_chunks(tok, W, dedup_across_chunks=False))
            ends.append(e)
            t_plan.append(ms)
        cnt = [[e[0]] + [e[q] - e[q - 1] for q in range(1, W)] for e in ends]        # cnt[r][q]: r sends to q
        for r, s in enumerate(shards):
            buf = torch.empty((max(ends[r][-1], 1), rec), dtype=torch.uint8, device="cuda")
            _, ms = timed(lambda: s.shard_gather_pack_range(0, ends[r][-1], buf[:ends[r][-1]]))
            sends.append(buf)
            t_pack.append(ms)
        for q, s in enumerate(shards):
            parts = []
            for r in range(W):
                o = sum(cnt[r][:q])
                parts.append(sends[r][o:o + cnt[r][q]])
            recv = torch.cat(parts).contiguous()                                      # the all-to-all, by hand
            b0, b1 = min(q * bper, B), min(q * bper + bper, B)

            def step():
                s.shard_gather_add_records(recv, 0, recv.shape[0])
                s.shard_gather_embed_range(tok, b0, b1, recv, out[b0 * T:b1 * T], wte=wte, wpe=wpe, out_is_slice=True)
            _, ms = timed(step)
            t_embed.append(ms)
        cur = [t_plan, t_pack, t_embed]
        best = cur if best is None else [[min(x, y) for x, y in zip(b, c_)] for b, c_ in zip(best, cur)]
    for r in range(W):
        res["ranks"].append({"rank": r, "plan_ms": best[0][r], "pack_ms": best[1][r], "embed_ms": best[2][r],
                             "send_records": int(sum(cnt[r])), "send_off_rank_bytes": int((sum(cnt[r]) - cnt[r][r]) * rec)})
    loc = [x["plan_ms"] + x["pack_ms"] + x["embed_ms"] for x in res["ranks"]]
    res["mode"] = "rows_dedup"
    res["local_ms_max"], res["local_ms_mean"] = max(loc), sum(loc) / len(loc)
    res["wire_bytes_all_ranks"] = sum(x["send_off_rank_bytes"] for x in res["ranks"])
    if a.check:
        full_t = SconeTable(3, N, d, a.format)
        full_t.index_build(keys, lens)
        full_t.fill_synthetic(7, 0.02 / 127)
        want = full_t.embed(tok, wte=wte, wpe=wpe).reshape(B * T, d)
        res["bit_identical_to_unsharded"] = bool(torch.equal(out, want))
    print(json.dumps(res))


def gather_rows_chunked(a, shards, tok, wte, wpe, out, res, keys, lens):
    """The pipelined all-gather form (ShardedEmbeddingCache._embed_gather_rows) with the C all-gathers done by hand: per rank,
    plan (one match + C claim passes), the C packs, and per chunk add_records + embed_range.  What a rank cannot hide behind
    its transfers: plan + first pack + 1/C of the collective + every chunk's reduction."""
    N, W, d, B, T, C = a.rows, a.world, a.dim, a.batch, a.seq, a.chunks
    rec = res["record_bytes"]
    per = (B + C - 1) // C
    best = None
    for rep in range(a.reps):
        ends, t_plan = [], []
        for s in shards:
            e, ms = timed(lambda: s.shard_gather_plan_chunks(tok, C, dedup_across_chunks=True))
            ends.append(e)
            t_plan.append(ms)
        mine = [[e[0]] + [e[c] - e[c - 1] for c in range(1, C)] for e in ends]
        maxc = [max(mine[r][c] for r in range(W)) for c in range(C)]
        base = [0]
        for c in range(C):
            base.append(base[-1] + W * maxc[c])
        full = torch.empty((max(base[-1], 1), rec), dtype=torch.uint8, device="cuda")
        t_pack = [0.0] * W
        for r, s in enumerate(shards):
            first = 0
            for c in range(C):
                if maxc[c]:
                    region = full[base[c] + r * maxc[c]:base[c] + (r + 1) * maxc[c]]      # = what the all-gather would deliver
                    _, ms = timed(lambda: s.shard_gather_pack_range(first, mine[r][c], region))
                    t_pack[r] += ms
                first += mine[r][c]
        records = full[:base[-1]]
        t_embed = [0.0] * W
        for q, s in enumerate(shards):
            for c in range(C):
                s0, s1 = min(c * per, B), min(c * per + per, B)

                def step():
                    if c == 0 or base[c + 1] > base[c]:
                        s.shard_gather_add_records(records, base[c], base[c + 1] - base[c])
                    if s1 > s0:
                        s.shard_gather_embed_range(tok, s0, s1, records, out, wte=wte, wpe=wpe)
                _, ms = timed(step)
                t_embed[q] += ms
        cur = [t_plan, t_pack, t_embed]
        best = cur if best is None else [[min(x, y) for x, y in zip(b, c_)] for b, c_ in zip(best, cur)]
        counts = [e[-1] for e in ends]
        chunk_bytes = [W * m * rec for m in maxc]
    for r in range(W):
        res["ranks"].append({"rank": r, "plan_ms": best[0][r], "pack_ms_all_chunks": best[1][r], "embed_ms_all_chunks": best[2][r],
                             "records_contributed": counts[r]})
    loc = [x["plan_ms"] + x["pack_ms_all_chunks"] + x["embed_ms_all_chunks"] for x in res["ranks"]]
    res["mode"], res["chunks"] = "gather_rows", C
    res["local_ms_max"], res["local_ms_mean"] = max(loc), sum(loc) / len(loc)
    res["all_gather_padded_bytes_per_chunk"] = chunk_bytes
    if a.wall:
        # rank 0's step as ShardedEmbeddingCache issues it, the other ranks' records already in place
        import time
        s = shards[0]

        def whole_step():
            e = s.shard_gather_plan_chunks(tok, C, dedup_across_chunks=True)          # synchronises (record counts)
            m = [e[0]] + [e[c] - e[c - 1] for c in range(1, C)]
            first = 0
            for c in range(C):
                if maxc[c]:
                    s.shard_gather_pack_range(first, m[c], full[base[c]:base[c] + maxc[c]])
                first += m[c]
            for c in range(C):
                s0, s1 = min(c * per, B), min(c * per + per, B)
                if c == 0 or base[c + 1] > base[c]:
                    s.shard_gather_add_records(records, base[c], base[c + 1] - base[c])
                if s1 > s0:
                    s.shard_gather_embed_range(tok, s0, s1, records, out, wte=wte, wpe=wpe)
        for _ in range(3):
            whole_step()
        torch.cuda.synchronize()
        walls = []
        for _ in range(10):
            t0 = time.perf_counter()
            whole_step()
            torch.cuda.synchronize()
            walls.append((time.perf_counter() - t0) * 1e3)
        res["rank0_whole_step_wall_ms"] = {"min": min(walls), "median": sorted(walls)[len(walls) // 2], "max": max(walls)}
        # ... and as the split-phase loop issues it (gather_rows_begin of batch b + 1 on a side stream, on the other plan
        # slot, while batch b is reduced on the main stream)
        side = torch.cuda.Stream(priority=-1 if a.side_high_priority else 0)
        fulls = [full, full.clone()]
        done = [None, None]

        def begin(slot):
            # (no wait for the caller's stream: the tokens of the next batch are ready long before -- gather_rows_begin(...,
            # tokens_ready=None); waiting for the current stream would put the plan BEHIND the reduction queued just before)
            if done[slot] is not None:
                side.wait_event(done[slot])
            with torch.cuda.stream(side):
                s.shard_select_slot(slot)
                e = s.shard_gather_plan_chunks(tok, C, dedup_across_chunks=True)
                m = [e[0]] + [e[c] - e[c - 1] for c in range(1, C)]
                first = 0
                for c in range(C):
                    if maxc[c]:
                        s.shard_gather_pack_range(first, m[c], fulls[slot][base[c]:base[c] + maxc[c]])
                    first += m[c]
                ready = torch.cuda.Event()
                ready.record(side)
            return slot, ready

        def finish(ticket):
            slot, ready = ticket
            cur = torch.cuda.current_stream()
            cur.wait_event(ready)
            s.shard_select_slot(slot)
            recs = fulls[slot][:base[-1]]
            for c in range(C):
                s0, s1 = min(c * per, B), min(c * per + per, B)
                if c == 0 or base[c + 1] > base[c]:
                    s.shard_gather_add_records(recs, base[c], base[c + 1] - base[c])
                if s1 > s0:
                    s.shard_gather_embed_range(tok, s0, s1, recs, out, wte=wte, wpe=wpe)
            done[slot] = torch.cuda.Event()
            done[slot].record(cur)

        def loop(n):
            slot = 0
            t = begin(slot)
            for i in range(n):
                finish(t)
                slot ^= 1
                t = begin(slot) if i + 1 < n else None
        loop(4)
        torch.cuda.synchronize()
        n = 20
        t0 = time.perf_counter()
        loop(n)
        torch.cuda.synchronize()
        res["rank0_split_phase_loop_ms_per_step"] = (time.perf_counter() - t0) * 1e3 / n
        res["side_stream_high_priority"] = bool(a.side_high_priority)
        s.shard_select_slot(0)
        if a.check:
            full_c = SconeTable(3, N, d, a.format)
            full_c.index_build(keys, lens)
            full_c.fill_synthetic(7, 0.02 / 127)
            res["split_phase_bit_identical_to_unsharded"] = bool(torch.equal(out, full_c.embed(tok, wte=wte, wpe=wpe).reshape(B * T, d)))
            del full_c
    if a.check:
        full_t = SconeTable(3, N, d, a.format)
        full_t.index_build(keys, lens)
        full_t.fill_synthetic(7, 0.02 / 127)
        want = full_t.embed(tok, wte=wte, wpe=wpe).reshape(B * T, d)
        res["bit_identical_to_unsharded"] = bool(torch.equal(out, want))
    print(json.dumps(res))


if __name__ == "__main__":
    main()
