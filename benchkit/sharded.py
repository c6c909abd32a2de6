"""N > 1: the `sharded` record of bench.py (the row-sharded path on the C5-shaped workload, one stage per exchange under the
watchdog).  Frozen since round 4."""
import time

from .common import HBM_PEAK_GBPS, XGMI_LINK_GBPS_PER_DIRECTION
from .roofline import workload_bytes
from .workloads import pinned_baseline


NCCL_HIGH_PRIORITY = [False]      # set by main() when the process group was created with a high-priority RCCL stream


def _rccl_version():
    try:
        import torch
        return ".".join(str(x) for x in torch.cuda.nccl.version())
    except Exception as e:
        return f"unknown ({e!r})"


def run_stages(rec, line, watchdog, budget, stage_limit, stages, n1_value=None, on_done=None):
    """The exchanges of the sharded record, one STAGE each: `stages` = [(name, callable -> dict)].  A stage runs under the
    watchdog with `stage_limit` seconds (less when the job's budget is nearly used up); one that raises is recorded as an
    error and the next one runs; one that HANGS (a collective that never completes) ends the job through the watchdog,
    which prints everything recorded before it -- hence the order of `stages`: the plainest collectives first.  Every
    finished stage is entered into `rec["exchanges"]` under the line's lock, with its speed-up over the N = 1 baseline."""
    for name, fn in stages:
        left = budget.remaining() - 15.0          # keep 15 s for the rest of the line
        if left < 10.0:
            with line.lock:
                rec["exchanges"][name] = {"skipped": f"time budget: {budget.remaining():.0f} s left"}
            continue
        watchdog.arm(f"sharded.exchanges.{name}", min(stage_limit, left))
        try:
            e = fn()
        except Exception as ex:                    # the record never takes the line down
            e = {"error": repr(ex)}
        watchdog.disarm()
        if n1_value and isinstance(e, dict) and e.get("tokens_per_s"):
            e["speedup_vs_n1_pinned_host"] = e["tokens_per_s"] / n1_value
        with line.lock:
            rec["exchanges"][name] = e
            if on_done is not None:
                on_done(name, e)


def sharded_record(args, dist, rank, world, backend, sync, rec, line, watchdog, budget):
    """N > 1: the row-sharded path on the C5-shaped workload -- INT4 d = 1024, `rows_per_rank` x N rows (1e9 at N = 8),
    replicated index built from keys generated on the GPU, every rank its own contiguous row range generated on its GPU,
    replicated head = the unigram rows, ONE 1M-token S_uniform batch that every rank passes in.  Six exchanges (round 4: the
    forms the measurements of rounds 2-3 chose; the chunked pipeline, records on the wire and the direct-mapped row map are
    gone from this record), in the order in which a hang costs least:
      world_sanity                      a 1 KB all-gather on every rank BEFORE the 100 GB build: a world that cannot even do
                                        that is reported in seconds
      gather_rows_split_phase           the serving loop (ShardedEmbeddingCache.gather_rows_begin / _finish; columns on the wire,
                                        match sharded over the ranks, 3 batches in flight): plan, pack and transfers of later
                                        steps run on side streams behind the reduction of this one.  Transport: three padded
                                        all_gather_into_tensor -- the plainest collective there is
      gather_rows_split_phase_p2p       the same, exact ranges over batch_isend_irecv (RCCL send / recv kernels)
      rows_slices_only                  the slice exchange alone: all_to_all_single of the distinct rows each slice needs, rank r
                                        reduces slice r and keeps it -- the ONLY form whose per-rank work shrinks with the world
      rows+all_gather                   the same + the all-gather of the finished fp16 vectors (the north-star's wording)
      gather_rows                       the one-call form of the all-gather exchange (nothing overlapped)
      gather_rows_split_phase_sdma      the serving loop over the copy engines (peer-mapped buffers, hipMemcpyAsync pushes,
                                        interprocess events): no transport kernel competes with the lookup grid.  Last: it is
                                        the newest transport, and what hangs here costs no other figure
    Before any of it rank 0 alone measures the N = 1 baseline (pinned host DRAM), so every exchange carries
    `speedup_vs_n1_pinned_host`.  Every exchange reads the handle's sticky status bits afterwards (`status_bits`: a row that
    never arrived, a token out of range): `exchanges_agree` needs them all zero.  Un-synchronised steps give ms/step; one
    instrumented step per one-call exchange (device synchronised between phases) gives the phase split.  `rec` is filled in
    place under `line.lock`."""
    import torch
    from scone_amd import synthetic as S
    from scone_amd.distributed import ShardedEmbeddingCache
    from scone_amd.hip_backend import format_code, row_bytes
    d, B, T = 1024, args.batch, args.seq
    cdev = "cuda" if backend == "nccl" else "cpu"
    # ---- world sanity: every rank contributes 1 KB, every rank checks what it got -- before anything expensive
    watchdog.arm("sharded.world_sanity", min(60.0, max(budget.remaining() - 30.0, 10.0)))
    t_s = time.perf_counter()
    mine = torch.full((256,), float(rank + 1), dtype=torch.float32, device=cdev)
    got = torch.empty(256 * world, dtype=torch.float32, device=cdev)
    dist.all_gather_into_tensor(got, mine)
    sane = bool(torch.equal(got.view(world, 256)[:, 0].cpu(), torch.arange(1, world + 1, dtype=torch.float32)))
    watchdog.disarm()
    line.set(rec, "world_sanity", {"all_gather_1KB_per_rank_ok": sane, "seconds": time.perf_counter() - t_s, "world_size": world})
    if not sane:
        raise RuntimeError("world sanity: a 1 KB all-gather returned the wrong ranks' data")
    # ---- the N = 1 baseline, rank 0 alone (the others wait in the first collective below)
    n1 = None
    if rank == 0:
        watchdog.arm("sharded.n1_pinned_host", min(150.0, max(budget.remaining() - 60.0, 10.0)))
        try:
            n1, _ = pinned_baseline(args, lambda: torch.cuda.synchronize(), zipf_too=False)
        except Exception as e:
            n1 = {"value": None, "error": repr(e)}
        watchdog.disarm()
        line.set(rec, "n1_pinned_host", n1)
    n1_value = (n1 or {}).get("value")
    # (ranks other than 0 enter this stage while rank 0 is still measuring its baseline: their limit includes that wait)
    watchdog.arm("sharded.build", min(200.0 + (150.0 if rank else 0.0), max(budget.remaining() - 30.0, 10.0)))
    free, total = torch.cuda.mem_get_info()
    fm = torch.tensor([float(free)], dtype=torch.float64, device=cdev)
    dist.all_reduce(fm, op=dist.ReduceOp.MIN)       # every rank must size the table the same way: the tightest GPU decides
    free = float(fm.item())
    per = args.sharded_rows_per_rank
    N = per * world
    cap = 64
    while cap < 2 * N:
        cap <<= 1
    # rows + scales, the index (16-B slots + bitmap), the claim tables (all-gather form: 4 B per local row; slice exchange: one
    # table per destination, 4 B x local rows x world -- 4 GB per rank at C5), buffers
    need = per * 544 + cap * 17 + per * 4 * (world + 1) + 12e9
    note = None
    if need > free:
        scale = max(0.05, (free - 12e9) / (need - 12e9))
        per = int(per * scale * 0.9)
        N = per * world
        note = f"rows per rank reduced to {per} ({free / 1e9:.0f} GB of HBM free)"
    vocab = S.StructuredVocab(N)
    t_build = time.perf_counter()
    cache = ShardedEmbeddingCache.from_synthetic(vocab, d, table_format="int4", seed=7, base_scale=0.02 / 127, rank=rank,
                                                 world=world, replicated_rows=S.GPT2_VOCAB, n_rows=N)
    torch.cuda.synchronize()
    t_build = time.perf_counter() - t_build
    tok = torch.from_numpy(S.stream_uniform_ids(vocab, None, B, T, 1234)).to("cuda", torch.int32)
    g = torch.Generator(device="cuda").manual_seed(5)
    wte = (torch.randn(S.GPT2_VOCAB, d, generator=g, device="cuda") * 0.02).half()
    wpe = (torch.randn(1024, d, generator=g, device="cuda") * 0.01).half()
    ntok = B * T
    # per-rank HBM bytes of a step (SURVEY 8d's accounting): the whole batch when every rank reduces it, rank 0's slice
    # when every rank reduces its own
    fmt = format_code("int4")
    alg_all, comp_all, sum_k, _, nr_all, nt_all = workload_bytes(cache.table, tok, fmt, d)
    bper = (B + world - 1) // world
    alg_sl, comp_sl, _, _, _, _ = workload_bytes(cache.table, tok[:bper], fmt, d)
    links = max(1, min(world - 1, 7))
    xgmi_peak = links * XGMI_LINK_GBPS_PER_DIRECTION
    with line.lock:
        rec.update({"workload": f"{N}-row int4 f-gram table d={d} row-sharded over {world} ranks ({per} rows = {per * 528 / 1e9:.1f} GB per rank; "
                                f"{args.sharded_rows_per_rank} rows per rank requested: N = 2 / 4 / 8 ranks hold {2 * per} / {4 * per} / {8 * per} rows), "
                                f"replicated {cap}-slot index, replicated head {S.GPT2_VOCAB} rows, structured vocabulary, S_uniform, "
                                f"{B}x{T} tokens/step (the same batch on every rank), whole [B,T,d] fp16 output on every rank",
                    "rows_total": N, "rows_per_rank": per, "mean_hits_per_token": sum_k / ntok,
                    "world_size": dist.get_world_size(), "device_count": torch.cuda.device_count(), "backend": backend,
                    "rccl_version": _rccl_version() if backend == "nccl" else None,
                    "rccl_high_priority_stream": bool(NCCL_HIGH_PRIORITY[0]) if backend == "nccl" else None,
                    "build_s": t_build, "note": note,
                    "xgmi_peak_GBps": xgmi_peak,
                    "xgmi_peak_kind": f"into one GPU: {links} links x {XGMI_LINK_GBPS_PER_DIRECTION} GB/s per direction "
                                      "(153.6 GB/s per link counting both directions)"})
    watchdog.disarm()
    checks = {}
    chunks = cache.gather_chunks

    def roofline_of(kw, ms_per_step, phases, wire_bytes):
        whole = kw["exchange"] == "gather_rows"
        alg, comp = (alg_all, comp_all) if whole else (alg_sl, comp_sl)
        if not whole and kw["gather_output"]:
            comp += (world - 1) * bper * T * d * 2                      # the other slices arrive and are written too
            alg += (world - 1) * bper * T * d * 2
        coll_ms = (phases.get("collective_ms", 0.0) + phases.get("gather_out_ms", 0.0)) if phases else None
        return {"bound": "hbm", "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                "per_rank_tokens_reduced": ntok if whole else bper * T,
                "per_rank_algorithmic_bytes": alg, "per_rank_compulsory_bytes": comp,
                "achieved": comp / (ms_per_step * 1e-3) / 1e9, "frac": comp / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                "frac_kind": "compulsory HBM bytes of one rank's step (distinct rows + wte rows + output + ids) / ms_per_step / 8 TB/s: "
                             "the whole exchange step, not one kernel",
                "algorithmic_frac": alg / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                "wire": {"bytes_received_rank0": wire_bytes, "collective_ms_instrumented": coll_ms,
                         "GBps": (wire_bytes / (coll_ms * 1e-3) / 1e9) if coll_ms else None,
                         "frac_of_xgmi_peak": (wire_bytes / (coll_ms * 1e-3) / 1e9 / xgmi_peak) if coll_ms else None}}

    # what one rank's step moves through ITS HBM at other world sizes (the same 1M-token batch): the forms in which every rank
    # reduces the whole batch do not get cheaper with more GPUs -- their >= 4x over the pinned-host baseline is HBM against
    # PCIe, not parallel speed-up; only the slice exchange divides the work
    by_world = {}
    for w in (2, 4, 8):
        bw = (B + w - 1) // w
        a_sl, c_sl, _, _, _, _ = workload_bytes(cache.table, tok[:bw], fmt, d)
        by_world[w] = {"whole_batch_on_every_rank": alg_all, "slice_only": a_sl, "slice_plus_gathered_output": a_sl + (w - 1) * bw * T * d * 2 * 2}

    def scaling_of(kw):
        whole = kw["exchange"] == "gather_rows"
        key = "whole_batch_on_every_rank" if whole else ("slice_plus_gathered_output" if kw["gather_output"] else "slice_only")
        return {"per_rank_hbm_bytes_vs_world": {str(w): by_world[w][key] for w in by_world},
                "scales_with_world": bool(not whole and not kw["gather_output"])}

    def status_bits():
        b = int(cache.table.status())
        tb = torch.tensor([float(b)], dtype=torch.float64, device=cdev)
        dist.all_reduce(tb, op=dist.ReduceOp.MAX)               # any rank's bits count
        return int(tb.item())

    def one_call(name, kw, transport):
        def fn():
            cache.gather_chunks = 1
            used = cache.set_gather_transport(transport)
            out = cache.embed_tokens(tok, wte=wte, wpe=wpe, **kw)                  # warm-up (allocations, RCCL channels)
            out = cache.embed_tokens(tok, wte=wte, wpe=wpe, **kw)
            sync()
            t0 = time.perf_counter()
            for _ in range(args.sharded_steps):
                out = cache.embed_tokens(tok, wte=wte, wpe=wpe, **kw)
            sync()
            dt = time.perf_counter() - t0
            tm = torch.tensor([dt], dtype=torch.float64, device=cdev)
            dist.all_reduce(tm, op=dist.ReduceOp.MAX)
            dt = float(tm.item())
            phases = cache.embed_tokens(tok, wte=wte, wpe=wpe, profile=True, **kw)[1]
            ph = torch.tensor([phases[k] for k in sorted(phases)], dtype=torch.float64, device=cdev)
            dist.all_reduce(ph, op=dist.ReduceOp.MAX)                                 # slowest rank per phase
            phm = {k: float(v) for k, v in zip(sorted(phases), ph.tolist()) if not k.startswith("bytes")}
            bits = status_bits()
            if kw["gather_output"]:
                checks[name] = float(out.float().abs().sum().item())
            ms = dt / args.sharded_steps * 1e3
            wire = int(phases.get("bytes_received", 0))
            return {"ms_per_step": ms, "tokens_per_s": ntok * args.sharded_steps / dt, "steps": args.sharded_steps,
                    "phase_ms_slowest_rank": phm, "wire_bytes_received_rank0": wire, "status_bits": bits,
                    "wire_format": ("columns: payload rows | scales | the senders' hash fragments"
                                    if kw["exchange"] == "gather_rows" else "records, one per distinct row and destination"),
                    "records_transport": (TRANSPORTS[used] if kw["exchange"] == "gather_rows" else "all_to_all_single"),
                    "roofline": roofline_of(kw, ms, phm, wire), **scaling_of(kw)}
        return name, fn

    def split_phase(name, transport, slots=3):
        def fn():
            cache.gather_chunks = 1
            used = cache.set_gather_transport(transport)
            prev_slots = cache.plan_slots
            cache.plan_slots = slots
            tickets = []

            def loop(n):
                o, nxt = None, 0
                for _ in range(min(slots - 1, n)):                          # slots - 1 batches ahead of the one being reduced
                    tickets.append(cache.gather_rows_begin(tok, tokens_ready=None))   # (the batch has been on the device since the build)
                    nxt += 1
                for i in range(n):
                    o = cache.gather_rows_finish(tickets.pop(0), wte=wte, wpe=wpe)    # queues the reduction of step i ...
                    if nxt < n:                                             # ... plan / pack / transfers of a later step overlap it
                        tickets.append(cache.gather_rows_begin(tok, tokens_ready=None))
                        nxt += 1
                return o
            reserve = None
            try:
                out = loop(3)
                sync()
                t0 = time.perf_counter()
                out = loop(args.sharded_steps)
                sync()
                dt = time.perf_counter() - t0
                # the same loop with the lookup kernel leaving R compute units to the transport kernels (scone_set_cu_reserve), its
                # reductions queued on the handle's CU-masked stream: measured with RCCL-shaped stand-in kernels on one GPU this
                # buys 10-18 % with the transfers in flight and costs 2-5 % without -- here it meets RCCL
                R = int(args.sharded_cu_reserve)
                if R > 0 and used != "sdma" and hasattr(cache.table, "set_cu_reserve"):
                    cache.table.set_cu_reserve(R)
                    try:
                        with torch.cuda.stream(cache.table.lookup_stream()):
                            loop(3)
                            sync()
                            t1 = time.perf_counter()
                            loop(args.sharded_steps)
                            sync()
                            dr = time.perf_counter() - t1
                        tr = torch.tensor([dr], dtype=torch.float64, device=cdev)
                        dist.all_reduce(tr, op=dist.ReduceOp.MAX)
                        reserve = {"compute_units_reserved": R, "ms_per_step": float(tr.item()) / args.sharded_steps * 1e3,
                                   "tokens_per_s": ntok * args.sharded_steps / float(tr.item())}
                    finally:
                        sync()
                        cache.table.set_cu_reserve(0)
            finally:
                # a stage that raised mid-loop must not leave its tickets open (the slots would refuse every later stage) nor
                # its settings behind
                for tk in tickets:
                    try:
                        cache.gather_rows_abandon(tk)
                    except Exception:
                        pass
                if tickets:
                    cache.reset_slots()
                cache.plan_slots = prev_slots
            tm = torch.tensor([dt], dtype=torch.float64, device=cdev)
            dist.all_reduce(tm, op=dist.ReduceOp.MAX)
            dt = float(tm.item())
            ms = dt / args.sharded_steps * 1e3
            bits = status_bits()
            kw = {"exchange": "gather_rows", "gather_output": True}
            checks[name] = float(out.float().abs().sum().item())
            return {"ms_per_step": ms, "tokens_per_s": ntok * args.sharded_steps / dt,
                    "steps": args.sharded_steps, "batches_in_flight": slots, "status_bits": bits,
                    "match": "sharded over the ranks + all-gather of the list records",
                    "wire_format": "columns: payload rows | scales | the senders' hash fragments",
                    "records_transport": TRANSPORTS[used], "transport_requested": transport,
                    "transport_fallback_reason": cache.transport_fallback_reason,
                    "with_cu_reserve": reserve, "sync_free_plan": dict(cache.sync_free_stats),
                    "roofline": roofline_of(kw, ms, None, wire_cols), **scaling_of(kw)}
        return name, fn

    TRANSPORTS = {"p2p": "batch_isend_irecv, exact ranges (RCCL send / recv kernels)", "all_gather": "all_gather_into_tensor, padded",
                  "sdma": "copy-engine pushes into peer-mapped buffers (hipMemcpyAsync), interprocess events, exact ranges"}
    # bytes the columns exchange puts into rank 0 (exact ranges): the distinct rows of the other ranks + their fragments
    # (counted once, from the match: the split-phase stages have no instrumented step)
    _, ids_all = cache.table.match_csr(tok)
    other = torch.unique(ids_all[(ids_all >= max(cache.row_end, S.GPT2_VOCAB)) | ((ids_all < cache.row_begin) & (ids_all >= S.GPT2_VOCAB))])
    wire_cols = int(other.numel()) * (512 + 16 + 32)             # payload + scales + 4 fragment slots of 8 B per row
    del ids_all, other
    whole = {"gather_output": True}
    stages = [split_phase("gather_rows_split_phase", "all_gather"),
              split_phase("gather_rows_split_phase_p2p", "p2p"),
              one_call("rows_slices_only", {"exchange": "rows", "gather_output": False}, "p2p"),
              one_call("rows+all_gather", {"exchange": "rows", **whole}, "p2p"),
              one_call("gather_rows", {"exchange": "gather_rows", **whole}, "all_gather"),
              split_phase("gather_rows_split_phase_sdma", "sdma")]
    with line.lock:
        rec["exchanges"] = {}
        rec["form_for_data_parallel_consumers"] = ("rows_slices_only: the only exchange whose per-rank HBM bytes fall with the world "
                                                   "size (`scales_with_world`); the gather_rows forms leave the whole output on "
                                                   "every rank and every rank pays for the whole batch")

    def on_done(name, e):                                  # (inside line.lock)
        ok = {k: v for k, v in rec["exchanges"].items() if isinstance(v, dict) and v.get("tokens_per_s")
              and k != "rows_slices_only" and not v.get("status_bits")}
        if ok:
            best = max(ok, key=lambda k: ok[k]["tokens_per_s"])
            rec["best_whole_output"] = {"exchange": best, "tokens_per_s": ok[best]["tokens_per_s"],
                                        "ms_per_step": ok[best]["ms_per_step"],
                                        "speedup_vs_n1_pinned_host": ok[best].get("speedup_vs_n1_pinned_host")}

    run_stages(rec, line, watchdog, budget, args.stage_limit, stages, n1_value, on_done)
    cache.gather_chunks = chunks
    with line.lock:
        if len(checks) >= 2:                             # all bit-identical to the unsharded table, hence to each other --
            bits = {k: v.get("status_bits") for k, v in rec["exchanges"].items() if isinstance(v, dict) and "status_bits" in v}
            rec["exchanges_agree"] = bool(len(set(checks.values())) == 1 and not any(bits.values()))   # and no status bit anywhere
            rec["exchanges_compared"] = sorted(checks)
            rec["status_bits"] = bits
        rec["n1_baseline"] = ("`n1_pinned_host` of THIS record (rank 0, same process, before the exchanges): one GPU cannot hold the "
                              "table, so its rows sit in pinned host DRAM and cross PCIe; '>= 4x at 8 GPUs vs 1 GPU' = "
                              "exchanges.<name>.speedup_vs_n1_pinned_host")
    watchdog.arm("sharded.close", min(45.0, max(budget.remaining() - 5.0, 5.0)))   # collective (two host barriers): a rank that
    cache.close()                                                                  # never arrives must not cost the rest of the budget
    watchdog.disarm()
    del cache, tok, wte, wpe
    torch.cuda.empty_cache()
    return rec
