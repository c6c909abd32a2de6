#!/usr/bin/env python3
"""Rank 0's local step of the C5 exchange at C5's TRUE scale, on one GPU: shard 0 of the 1e9-row INT4 d = 1024 table
(125M rows = 66 GB) with the full replicated 1e9-key index (2^31 slots, 34 GB) and the replicated unigram head, a
1M-token S_uniform batch, `exchange="gather_rows"` in one piece.  The records of the other seven ranks cannot be produced
here (their rows are not on this GPU), but everything rank 0 does with them can be timed: they are laid out in the
all-gather's receive buffer with the RIGHT row ids (every distinct non-head row of ranks 1..7 that the batch references,
grouped by owner, padded like the all-gather pads) and zero payloads -- the lookup reads the same bytes from the same
places, only the values differ.

Times rank 0's step (a) issued back to back on one stream and (b) as the split-phase loop issues it (gather_rows_begin of
the next batch(es) on a side stream / other plan slots behind gather_rows_finish of this one) for several FLOWS, alternating
in one process (a box runs the same flow 2-4 % apart from one process to the next; medians over --rounds are printed):
the round-2 flow (every rank matches the whole batch, records on the wire), + the match sharded over the ranks (the other
ranks' list records arrive by a device copy), + columns on the wire (round 3: payload rows | scales | sender-built hash
fragments; the other ranks' fragments are real, built from their row ids), three batches in flight, and a variant that was
measured and not kept (post stream; the direct-mapped row map of round 3 is gone from the library).  All flows must give the
same output checksum.
tools/shard_emulate.py does the same for ALL eight ranks of a 100M-row table, with real payloads and the bit-exactness check.

Round 4, `--transport-standin`: the figures above assume that the transfers cost the step nothing ("the other ranks' records
already in place").  On real links they are RCCL kernels -- a few channels per peer, one 256-512-thread workgroup each --
that need wave slots and memory bandwidth WHILE the lookup grid holds every slot of the chip.  With this switch every
transfer of rank 0's step is really executed, concurrently with the reduction, by kernels of that shape
(tools/standin/transport_standin.hip): the all-gather of the list records (29 MB in, 4 MB out to each of 7 peers), and the
three column exchanges (258 MB in from 7 staging buffers, rank 0's own ~35 MB out to 7 peer buffers) -- local HBM reads and
writes where xGMI would carry one side, so HBM contention is over-, not under-stated.  Every flow is timed with and
without that traffic, alternating in one process, for every `--cu-reserve` R (scone_set_cu_reserve: the lookup kernel leaves R
compute units to the transport kernels).
"""
import argparse
import contextlib
import ctypes
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from scone_amd import synthetic as S
from scone_amd.distributed import ShardedEmbeddingCache


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=1_000_000_000)
    ap.add_argument("--world", type=int, default=8)
    ap.add_argument("--batch", type=int, default=2048)
    ap.add_argument("--seq", type=int, default=512)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--padded", action="store_true", help="the padded all-gather's receive layout instead of exact ranges")
    ap.add_argument("--variants", default="0,1,2,3", help="which flows to time (0 = round 2, 1 = + sharded match, 2 = + columns on the "
                    "wire (round 3), 3 = three batches in flight, 4 = 1 + post stream)")
    ap.add_argument("--rounds", type=int, default=5, help="alternating rounds over the chosen flows (medians are reported)")
    ap.add_argument("--one-only", action="store_true", help="time the one-stream step only (clean per-kernel times under a profiler)")
    ap.add_argument("--split-only", action="store_true", help="time the split-phase loop only (for a kernel profile of it)")
    ap.add_argument("--transport-standin", nargs="?", const="kernel", default=None, choices=["kernel", "sdma"],
                    help="every flow also WITH its transfers really executed, concurrently with the reduction (columns flows: "
                         "variants 2, 3) -- kernel: RCCL-shaped copy kernels (a few workgroups per peer); sdma: the copy engines "
                         "(hipMemcpyAsync without compute units, one stream per peer and direction: what gather_transport='sdma' does)")
    ap.add_argument("--standin-directions", default="both", choices=["both", "out", "out-read"], help="out-read: as out, but the "
                    "send kernels only READ (their writes would leave over xGMI): the lower bracket of what a send costs this GPU; "
                    "kernel stand-in only.  " "both: what arrives is copied staging -> "
                    "receive buffer by this GPU too (RCCL's FIFO protocol; doubles the local work); out: only what rank 0 SENDS is "
                    "moved (peers write straight into rank 0's buffers: RCCL direct / the sdma transport -- the incoming bytes cost "
                    "this GPU HBM write bandwidth only, which the emulation then leaves out)")
    ap.add_argument("--channels", type=int, default=2, help="stand-in: workgroups per peer and direction for the large segments")
    ap.add_argument("--threads", type=int, default=256, help="stand-in: threads per workgroup (256 or 512)")
    ap.add_argument("--cu-reserve", default="0", help="comma list of R: compute units the lookup kernel leaves free")
    ap.add_argument("--reserve-mode", default="direct", choices=["direct", "hop"], help="direct: the loop's main stream IS the "
                    "handle's CU-masked stream (scone_lookup_stream); hop: the library moves each lookup there between two events")
    a = ap.parse_args()
    N, W, d, B, T = a.rows, a.world, 1024, a.batch, a.seq
    vocab = S.StructuredVocab(N)
    t0 = time.perf_counter()
    cache = ShardedEmbeddingCache.from_synthetic(vocab, d, table_format="int4", seed=7, base_scale=0.02 / 127, rank=0, world=W,
                                                 replicated_rows=S.GPT2_VOCAB, n_rows=N)
    torch.cuda.synchronize()
    build_s = time.perf_counter() - t0
    s = cache.table
    tok = torch.from_numpy(S.stream_uniform_ids(vocab, None, B, T, 1234)).to("cuda", torch.int32)
    g = torch.Generator(device="cuda").manual_seed(5)
    wte = (torch.randn(S.GPT2_VOCAB, d, generator=g, device="cuda") * 0.02).half()
    wpe = (torch.randn(1024, d, generator=g, device="cuda") * 0.01).half()
    out = torch.empty((B * T, d), dtype=torch.float16, device="cuda")
    rec = s.shard_record_bytes()
    per = N // W
    # what the other ranks contribute: the distinct rows outside my range (and outside the head) the batch references
    _, ids = s.match_csr(tok)
    other = torch.unique(ids[ids >= cache.row_end].to(torch.int64))
    owner = torch.div((other + 1) * W - 1, N, rounding_mode="floor")
    counts = [int((owner == r).sum().item()) for r in range(W)]
    mine = s.shard_gather_plan_chunks(tok, 1)[0]
    counts[0] = mine
    # receive-buffer layout: exact point-to-point ranges back to back (the default transport), or -- --padded -- every
    # contribution padded to the largest as all_gather_into_tensor needs
    size = [max(counts)] * W if a.padded else counts
    offs = [0]
    for r in range(W):
        offs.append(offs[-1] + size[r])
    total = offs[-1]
    full = torch.zeros((total, rec), dtype=torch.uint8, device="cuda")
    hdr = full.view(torch.int32).view(total, rec // 4)
    hdr[:, rec // 4 - 2] = -1                                            # every record is padding ...
    hdr[:, rec // 4 - 1] = -1
    other_by_rank = {}
    for r in range(1, W):
        rows_r = other[owner == r]
        other_by_rank[r] = rows_r.to(torch.int32)
        hdr[offs[r]:offs[r] + rows_r.numel(), rec // 4 - 2] = rows_r.to(torch.int32)     # ... except the real ones: row id
    del ids, other, owner
    fulls = [full, full.clone(), full.clone()]
    maxc = size[0]
    # ---- the plan's match sharded over the ranks (round 3): rank 0 matches only ITS slice (1/W of the sequences); the
    # list records of the other slices arrive by all-gather -- emulated by a device copy out of a pre-computed set of lists
    # (reads AND writes the 28 MB where a real receive only writes them: the emulation charges a little more than RCCL would)
    wd = s.ell_width()
    bper = (B + W - 1) // W
    ell_src = torch.empty((W * bper * T, wd), dtype=torch.int32, device="cuda")
    s.shard_gather_match(tok, 0, B, ell_src)                               # what the gathered lists will hold
    ells = [ell_src.clone(), ell_src.clone(), ell_src.clone()]
    send = torch.empty((bper * T, wd), dtype=torch.int32, device="cuda")
    results = {}
    flows = (("round2: every rank matches the whole batch", "hash", False, False, 2),
             ("match sharded over the ranks", "hash", True, False, 2),
             ("round3: sharded match + columns on the wire (payload rows | scales | the senders' hash fragments: no indexing pass)",
              "cols", True, False, 2),
             ("round3, three batches in flight", "cols", True, False, 3),
             ("sharded match + records indexed / lists remapped on a post stream", "hash", True, True, 2))
    # ---- columns on the wire: receive buffers per slot, the other ranks' columns synthesised once (the batch never changes):
    # zero payloads and scales, REAL hash fragments of their row ids
    from scone_amd.hip_backend import SconeTable
    pbytes, sbytes, nh = s.payload_bytes(), s.scale_bytes(), S.GPT2_VOCAB
    cslots = [SconeTable.cols_frag_slots(c) for c in counts]
    crec = [sum(counts[:r]) for r in range(W)]
    cfoff = [sum(cslots[:r]) for r in range(W)]
    ctotal = sum(counts)
    c_rows, c_scales, c_frags = [], [], []
    for k in range(3):
        c_rows.append(torch.zeros((ctotal, pbytes), dtype=torch.uint8, device="cuda"))
        sc = torch.zeros((nh + ctotal, sbytes), dtype=torch.uint8, device="cuda")
        s.shard_head_scales_into(sc)
        c_scales.append(sc)
        fr = torch.zeros(sum(cslots), dtype=torch.int64, device="cuda")
        for r in range(1, W):
            s.shard_cols_build_frag(other_by_rank[r], fr[cfoff[r]:cfoff[r] + cslots[r]])
        c_frags.append(fr)
    side, post = torch.cuda.Stream(), torch.cuda.Stream()

    # ---- stand-in transport (round 4): what the other ranks send sits in staging buffers (zero payloads and scales, their REAL
    # fragments and list records), what rank 0 sends goes to seven "peer" buffers; RCCL-shaped kernels move both
    standin = None
    if a.transport_standin:
        lib = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "standin", "libtransport_standin.so"))
        lib.standin_launch.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int,
                                       ctypes.c_void_p]
        src_rows = torch.zeros((ctotal, pbytes), dtype=torch.uint8, device="cuda")
        src_scales = torch.zeros((ctotal, sbytes), dtype=torch.uint8, device="cuda")
        src_frags = c_frags[0].clone()
        peer_rows = [torch.empty((counts[0], pbytes), dtype=torch.uint8, device="cuda") for _ in range(W - 1)]
        peer_scales = [torch.empty((counts[0], sbytes), dtype=torch.uint8, device="cuda") for _ in range(W - 1)]
        peer_frags = [torch.empty(cslots[0], dtype=torch.int64, device="cuda") for _ in range(W - 1)]
        peer_ell = [torch.empty((bper * T, wd), dtype=torch.int32, device="cuda") for _ in range(W - 1)]

        push_streams = [torch.cuda.Stream() for _ in range(2 * (W - 1))]

        def group_sdma(segs):
            """The same segments on the copy engines: segment i on stream i mod 14 (7 peers x 2 directions), no kernel at all."""
            cur = torch.cuda.current_stream()
            ready = torch.cuda.Event()
            ready.record(cur)
            for i, (x, y, _) in enumerate(segs):
                if not x.numel():
                    continue
                ps = push_streams[i % len(push_streams)]
                ps.wait_event(ready)
                with torch.cuda.stream(ps):
                    s.ipc_push(y.data_ptr(), x.data_ptr(), x.numel() * x.element_size(), True)
                    e = torch.cuda.Event()
                    e.record(ps)
                cur.wait_event(e)

        def group(segs):
            """One RCCL group = one launch: segs = [(src tensor, dst tensor, channels)], bytes taken from src."""
            if a.transport_standin == "sdma":
                return group_sdma(segs)
            segs = [(x, y, c) for x, y, c in segs if x.numel()]
            n = len(segs)
            arr_p, arr_u, arr_i = ctypes.c_void_p * n, ctypes.c_ulonglong * n, ctypes.c_int * n
            src = arr_p(*[x.data_ptr() for x, _, _ in segs])
            dst = arr_p(*[(None if a.standin_directions == "out-read" else y.data_ptr()) for _, y, _ in segs])
            nb = arr_u(*[x.numel() * x.element_size() for x, _, _ in segs])
            ch = arr_i(*[c for _, _, c in segs])
            rc = lib.standin_launch(n, src, dst, nb, ch, a.threads, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
            assert rc == 0, rc

        def move_lists(slot):
            """all-gather of the list records: the seven other slices in, my slice out to seven peers."""
            if a.standin_directions != "both":           # the peers' parts: in place by a plain device copy, as without traffic
                ells[slot][bper * T:].copy_(ell_src[bper * T:])
                return group([(send, peer_ell[r - 1], a.channels) for r in range(1, W)])
            segs = [(ell_src[r * bper * T:(r + 1) * bper * T], ells[slot][r * bper * T:(r + 1) * bper * T], a.channels) for r in range(1, W)]
            segs += [(send, peer_ell[r - 1], a.channels) for r in range(1, W)]
            group(segs)

        def move_columns(slot, n):
            """the three column exchanges, one group each (as three batch_isend_irecv calls would be)."""
            if a.standin_directions != "both":           # (the other ranks' columns are in place since the set-up)
                group([(c_rows[slot][:n], peer_rows[r - 1], a.channels) for r in range(1, W)])
                group([(c_scales[slot][nh:nh + n], peer_scales[r - 1], 1) for r in range(1, W)])
                return group([(c_frags[slot][:cslots[0]], peer_frags[r - 1], 1) for r in range(1, W)])
            group([(src_rows[crec[r]:crec[r] + counts[r]], c_rows[slot][crec[r]:crec[r] + counts[r]], a.channels) for r in range(1, W)]
                  + [(c_rows[slot][:n], peer_rows[r - 1], a.channels) for r in range(1, W)])
            group([(src_scales[crec[r]:crec[r] + counts[r]], c_scales[slot][nh + crec[r]:nh + crec[r] + counts[r]], 1) for r in range(1, W)]
                  + [(c_scales[slot][nh:nh + n], peer_scales[r - 1], 1) for r in range(1, W)])
            group([(src_frags[cfoff[r]:cfoff[r] + cslots[r]], c_frags[slot][cfoff[r]:cfoff[r] + cslots[r]], 1) for r in range(1, W)]
                  + [(c_frags[slot][:cslots[0]], peer_frags[r - 1], 1) for r in range(1, W)])
        standin = (move_lists, move_columns)

    def make(variant, row_map, sharded_match, post_stream, slots, traffic=False):
        """(one-stream step, split-phase loop) of one flow as closures; the row map's form is read from the environment
        when an exchange starts, so it is set before every call.  traffic: the transfers are executed (stand-in kernels)."""
        def begin(slot):
            s.shard_select_slot(slot)
            if sharded_match:
                s.shard_gather_match(tok, 0, min(bper, B), send)           # my slice: sequences [0, bper)
                ells[slot][:bper * T].copy_(send)                          # (the all-gather's output: my part ...
                if traffic:
                    standin[0](slot)
                else:
                    # ... and the seven other ranks' parts: a device copy (28 MB read + 28 MB written where a receive only
                    # writes; it must be a fresh copy, the reduction rewrites the lists in place)
                    ells[slot][bper * T:].copy_(ell_src[bper * T:])
                n = s.shard_gather_plan_ell(ells[slot], B, T, 1)[0]
            else:
                n = s.shard_gather_plan_chunks(tok, 1)[0]
            if row_map == "cols":
                assert n == counts[0]
                s.shard_cols_pack(0, n, c_rows[slot][:n], c_scales[slot][nh:nh + n], c_frags[slot][:cslots[0]])
                if traffic:
                    standin[1](slot, n)
            else:
                s.shard_gather_pack_range(0, n, fulls[slot][:maxc])

        def index_remap(slot):
            s.shard_select_slot(slot)
            s.shard_gather_add_records(fulls[slot][:total], 0, total)
            s.shard_gather_remap_range(0, B)

        def finish(slot, indexed=False):
            s.shard_select_slot(slot)
            if row_map == "cols":
                s.shard_cols_embed(tok, 0, B, c_rows[slot], ctotal, c_scales[slot], c_frags[slot], cfoff, cslots, crec, out,
                                   wte=wte, wpe=wpe)
                return
            if not indexed:
                s.shard_gather_add_records(fulls[slot][:total], 0, total)
            s.shard_gather_embed_range(tok, 0, B, fulls[slot][:total], out, wte=wte, wpe=wpe)

        def one_stream(n):
            for _ in range(n):
                begin(0)
                finish(0)

        done = [None] * 3

        def begin_side(slot):
            if done[slot] is not None:
                side.wait_event(done[slot])
            with torch.cuda.stream(side):
                begin(slot)
                ready = torch.cuda.Event()
                ready.record(side)
            if post_stream:                      # (the transfers would be waited for here: the side stream plans on)
                post.wait_event(ready)
                with torch.cuda.stream(post):
                    index_remap(slot)
                    ready = torch.cuda.Event()
                    ready.record(post)
            return slot, ready

        def finish_main(ticket):
            slot, ready = ticket
            cur = torch.cuda.current_stream()
            cur.wait_event(ready)
            finish(slot, indexed=post_stream)
            done[slot] = torch.cuda.Event()
            done[slot].record(cur)

        def loop(n):
            for k in range(3):
                done[k] = None
            nxt, q = 0, []
            for _ in range(min(slots - 1, n)):   # slots - 1 batches ahead of the one being reduced
                q.append(begin_side(nxt % slots))
                nxt += 1
            for i in range(n):
                finish_main(q.pop(0))
                if nxt < n:
                    q.append(begin_side(nxt % slots))
                    nxt += 1
            torch.cuda.synchronize()
            s.shard_select_slot(0)
        return one_stream, loop

    chosen = [flows[int(i)] for i in a.variants.split(",")]
    reserves = [int(x) for x in a.cu_reserve.split(",")]
    fns = {}
    for f in chosen:
        for R in reserves:
            tag = f[0] if R == 0 else f"{f[0]} | {R} CUs reserved"
            fns[tag] = (R,) + make(*f)
            if a.transport_standin and f[1] == "cols":
                fns[tag + " | transfers in flight"] = (R,) + make(*f, traffic=True)
    samples = {name: {"one": [], "split": []} for name in fns}
    checks = {}
    def on_main(R):
        """the stream the loop's reductions are queued on: the handle's masked stream (no cross-stream events), or the default"""
        s.set_cu_reserve(R)
        return torch.cuda.stream(s.lookup_stream()) if (R and a.reserve_mode == "direct") else contextlib.nullcontext()

    for name, (R, one, loop) in fns.items():     # warm-up: allocations, the maps of every slot
        if name.endswith("transfers in flight") and a.standin_directions == "both":  # the stand-in must really deliver: wipe what it is to bring, then check
            for k in range(3):
                c_frags[k][cslots[0]:].zero_()
                ells[k][bper * T:].zero_()
        torch.cuda.synchronize()
        with on_main(R):
            one(3)
            torch.cuda.synchronize()
            assert s.status() == 0, "a referenced row is missing from the synthesised records"
            if not a.one_only:
                loop(4)
        torch.cuda.synchronize()
        assert s.status() == 0
        checks[name] = out.float().abs().sum().item()
        if name.endswith("transfers in flight") and a.standin_directions == "both":  # slots this flow did not use stay wiped: put the other ranks' parts back
            torch.cuda.synchronize()
            for k in range(3):
                c_frags[k][cslots[0]:].copy_(src_frags[cslots[0]:])
    for rnd in range(a.rounds):                  # alternating: the same box runs 2 % apart from one minute to the next
        for name, (R, one, loop) in fns.items():
            torch.cuda.synchronize()
            with on_main(R):
                if not a.split_only:
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    one(a.steps)
                    torch.cuda.synchronize()
                    samples[name]["one"].append((time.perf_counter() - t0) * 1e3 / a.steps)
                if a.one_only:
                    continue
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                loop(a.steps)
                samples[name]["split"].append((time.perf_counter() - t0) * 1e3 / a.steps)
    assert s.status() == 0
    s.set_cu_reserve(0)

    def med(v):
        v = sorted(v)
        return v[len(v) // 2] if v else None
    for name in fns:
        results[name] = {"rank0_step_one_stream_ms": med(samples[name]["one"]), "rank0_step_split_phase_loop_ms": med(samples[name]["split"]),
                         "split_phase_loop_ms_rounds": [round(x, 4) for x in samples[name]["split"]],
                         "one_stream_ms_rounds": [round(x, 4) for x in samples[name]["one"]], "output_checksum": checks[name]}
    sums = {v["output_checksum"] for v in results.values()}
    print(json.dumps({"rows": N, "world": W, "rank": 0, "tokens": B * T, "build_s": build_s, "records_per_rank": counts,
                      "layout": "padded to the largest contribution" if a.padded else "exact ranges",
                      "bytes_into_rank0": int((total - size[0]) * rec), "record_bytes": rec,
                      "list_record_bytes_gathered_into_rank0": int((W - 1) * bper * T * wd * 4),
                      "variants": results, "all_variants_same_output": len(sums) == 1,
                      "reserve_mode": a.reserve_mode,
                      "transport_standin": None if not a.transport_standin else {
                          "directions": a.standin_directions,
                          "kind": ("RCCL-shaped copy kernels" if a.transport_standin == "kernel" else
                                   "copy engines (hipMemcpyAsync, hipMemcpyDeviceToDeviceNoCU), 14 streams"),
                          "channels_per_peer_and_direction": a.channels, "threads_per_workgroup": a.threads,
                          "bytes_in": int((ctotal - counts[0]) * (pbytes + sbytes) + (sum(cslots) - cslots[0]) * 8 + (W - 1) * bper * T * wd * 4),
                          "bytes_out_per_peer": int(counts[0] * (pbytes + sbytes) + cslots[0] * 8 + bper * T * wd * 4),
                          "groups_per_step": 4, "workgroups_in_the_rows_group": 2 * (W - 1) * a.channels},
                      "note": "other ranks' records carry the right row ids and zero payloads; the all-gather of list records is "
                              "emulated by a device copy"}))


if __name__ == "__main__":
    main()
