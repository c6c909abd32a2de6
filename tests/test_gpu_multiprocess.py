"""GPU, several PROCESSES sharing the one card: the row-sharded path with real device shards
(``hip_backend.SconeTable`` with row_begin/row_end), real process separation and a real
``torch.distributed`` group.  RCCL refuses two ranks on one device, so the group is gloo and
``scone_amd.distributed`` stages the collectives through the host; everything else -- plan,
pack, the fused kernel reading the received records in place, finalise, the output all-gather
-- is the code an 8-GPU RCCL run executes.  Also launches ``bench.py --gpus 2`` the way the
driver does (``python -m torch.distributed.run``) and checks its JSON line.

At most 3 ranks touch the GPU at once (the box allows 6)."""

import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _problem(fmt, d, max_n, seed=5):
    rng = np.random.default_rng(seed)
    vocab, n = 23, 700
    lens = rng.integers(1, max_n + 1, size=n).astype(np.uint8)
    keys = rng.integers(0, vocab, size=(n, max_n)).astype(np.uint32)
    keys[np.arange(max_n)[None, :] >= lens[:, None]] = 0
    table = rng.standard_normal((n, d)).astype(np.float32)
    B, T = 5, 33                                   # B not a multiple of the world size: padded tail slice
    tok = rng.integers(0, vocab + 1, size=(B, T))  # id `vocab` never matches
    wte = (rng.standard_normal((vocab + 1, d)) * 0.1).astype(np.float32)
    wpe = (rng.standard_normal((T, d)) * 0.1).astype(np.float32)
    return keys, lens, table, tok, wte, wpe


def _worker(rank, world, port, fmt, d, max_n, exchange, head, q):
    try:
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        import torch.distributed as dist
        dist.init_process_group("gloo", rank=rank, world_size=world)
        torch.cuda.set_device(0)
        from scone_amd import EmbeddingCache, NGramExtractor
        from scone_amd.distributed import ShardedEmbeddingCache
        keys, lens, table, tok, wte, wpe = _problem(fmt, d, max_n)
        ex = NGramExtractor.from_arrays(keys, lens, max_n=max_n)
        exchange, _, rest = exchange.partition(":")               # "gather_rows:all_gather" = the padded all-gather transport,
        transport, _, rest = rest.partition(":")                  # "...:sm" = the plan's match sharded over the ranks,
        sm, _, c1 = rest.partition(":")                           # "...:c1" = one piece (columns on the wire); else the legacy 4-chunk pipeline (records)
        sabotaged = transport == "sdma_sabotaged"    # the transport's self-test finds rank 1's bytes missing: every rank must
        if sabotaged:                                # fall back to p2p, say why, and still return the right rows
            os.environ["SCONE_SDMA_SELF_TEST_SKIP_PUSH_OF_RANK"] = "1"
            transport = "sdma"
        sh = ShardedEmbeddingCache(ex, d, table_format=fmt, rank=rank, world=world, replicated_rows=head,
                                   gather_transport=transport or "p2p", shard_match=True if sm == "sm" else "auto",
                                   gather_chunks=1 if c1 else 4)
        if sabotaged:
            assert sh.gather_transport == "p2p" and "did not arrive" in (sh.transport_fallback_reason or ""), sh.transport_fallback_reason
        elif transport == "sdma":                    # the copy-engine transport must really be in use, not its fallback
            assert sh.gather_transport == "sdma" and sh.transport_fallback_reason is None, sh.transport_fallback_reason
        sh.load_rows(torch.from_numpy(table), 0)
        wte_d, wpe_d = torch.from_numpy(wte).half().cuda(), torch.from_numpy(wpe).half().cuda()
        got = sh.embed_tokens(torch.from_numpy(tok), wte=wte_d, wpe=wpe_d, exchange=exchange, check=True)
        # the unsharded table on the same GPU
        full = EmbeddingCache(ex, d, table_format=fmt)
        full.cache_embeddings(list(range(len(lens))), torch.from_numpy(table), verbose=False)
        ref = full.embed_tokens(torch.from_numpy(tok), wte=wte_d, wpe=wpe_d)
        same = bool(torch.equal(got, ref))
        err = float((got.float() - ref.float()).abs().max() / ref.float().abs().max())
        sl = sh.embed_tokens(torch.from_numpy(tok), wte=wte_d, wpe=wpe_d, exchange=exchange, gather_output=False)
        q.put((rank, same, err, tuple(got.shape), tuple(sl.shape)))
        dist.barrier()
        sh.close()
        dist.destroy_process_group()
    except Exception as e:                          # surface the failure in the parent
        import traceback
        q.put((rank, False, repr(e) + traceback.format_exc(), None, None))


@pytest.mark.parametrize("fmt,d,max_n,world,exchange,head", [("int8", 768, 3, 2, "rows", 0), ("int4", 1024, 4, 3, "rows", 100),
                                                             ("int8", 768, 3, 3, "gather_rows", 0),
                                                             ("int4", 1024, 4, 2, "gather_rows", 100),
                                                             ("int4", 1024, 3, 3, "gather_rows:all_gather", 100),
                                                             ("int4", 1024, 3, 3, "gather_rows:p2p:sm", 100),
                                                             ("int8", 768, 4, 2, "gather_rows:all_gather:sm", 0),
                                                             ("fp16", 768, 3, 3, "rows:p2p:sm", 100),
                                                             ("int4", 1024, 3, 3, "gather_rows:p2p:sm:c1", 100),
                                                             ("fp16", 768, 4, 2, "gather_rows:all_gather::c1", 0),
                                                             ("int8", 768, 3, 3, "gather_rows:all_gather:sm:c1", 37),
                                                             ("int4", 1024, 3, 3, "gather_rows:sdma:sm:c1", 100),
                                                             ("int8", 768, 3, 2, "gather_rows:sdma::c1", 0),
                                                             ("int8", 768, 3, 3, "gather_rows:sdma_sabotaged::c1", 0),
                                                             ("int8", 768, 3, 2, "partial_sums", 0)])
def test_sharded_cache_across_processes_on_one_gpu(fmt, d, max_n, world, exchange, head):
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, fmt, d, max_n, exchange, head, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        if p.is_alive():                                # (a rank left waiting for a peer that failed)
            p.terminate()
    for rank, same, err, shape, sl_shape in results:
        assert shape is not None, f"rank {rank} failed: {err}"
        assert shape == (5, 33, d)
        if exchange.partition(":")[0] in ("rows", "gather_rows"):
            assert same, f"rank {rank}: row exchange must be bit-identical to the unsharded table (rel err {err})"
        else:
            assert err < 1e-3, (rank, err)          # fp32 partial sums are added in shard order, not list order


def _worker_split_phase(rank, world, port, fmt, d, max_n, head, chunks, q, slots=2, transport="p2p"):
    try:
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        import torch.distributed as dist
        dist.init_process_group("gloo", rank=rank, world_size=world)
        torch.cuda.set_device(0)
        from scone_amd import EmbeddingCache, NGramExtractor
        from scone_amd.distributed import ShardedEmbeddingCache
        keys, lens, table, tok0, wte, wpe = _problem(fmt, d, max_n)
        rng = np.random.default_rng(99)
        batches = [tok0] + [rng.integers(0, 24, size=tok0.shape) for _ in range(4)]
        ex = NGramExtractor.from_arrays(keys, lens, max_n=max_n)
        sh = ShardedEmbeddingCache(ex, d, table_format=fmt, rank=rank, world=world, replicated_rows=head, gather_chunks=chunks,
                                   shard_match=chunks == 1, plan_slots=slots,   # (one of the cases: match sharded over the ranks)
                                   gather_transport=transport)
        if transport == "sdma":
            assert sh.gather_transport == "sdma" and sh.transport_fallback_reason is None, sh.transport_fallback_reason
        sh.load_rows(torch.from_numpy(table), 0)
        wte_d, wpe_d = torch.from_numpy(wte).half().cuda(), torch.from_numpy(wpe).half().cuda()
        full = EmbeddingCache(ex, d, table_format=fmt)
        full.cache_embeddings(list(range(len(lens))), torch.from_numpy(table), verbose=False)
        outs, tickets, nxt = [], [], 0
        for _ in range(slots - 1):                                      # slots - 1 batches ahead of the one being reduced
            tickets.append(sh.gather_rows_begin(torch.from_numpy(batches[nxt])))
            nxt += 1
        for i in range(len(batches)):                                   # begin(i + slots - 1) is queued on the side streams
            outs.append(sh.gather_rows_finish(tickets.pop(0), wte=wte_d, wpe=wpe_d, check=i == 2))   # behind finish(i) on the main one
            if nxt < len(batches):
                tickets.append(sh.gather_rows_begin(torch.from_numpy(batches[nxt])))
                nxt += 1
        torch.cuda.synchronize()
        same = all(bool(torch.equal(o, full.embed_tokens(torch.from_numpy(b), wte=wte_d, wpe=wpe_d))) for o, b in zip(outs, batches))
        # device tokens: produced on the caller's stream (default: the side stream waits for that stream), or with the
        # producer's event / nothing to wait for
        dev = [torch.from_numpy(b).cuda() for b in batches[:3]]
        ev = torch.cuda.Event()
        ev.record()
        outs2 = []
        ticket = sh.gather_rows_begin(dev[0])
        for i, ready in enumerate((ev, None, "current")):
            outs2.append(sh.gather_rows_finish(ticket, wte=wte_d, wpe=wpe_d))
            ticket = sh.gather_rows_begin(dev[i + 1], tokens_ready=ready) if i + 1 < 3 else None
        torch.cuda.synchronize()
        same = same and all(bool(torch.equal(a, b)) for a, b in zip(outs2, outs[:3]))
        # the one-call form still works afterwards (slot 0, current stream)
        again = bool(torch.equal(sh.embed_tokens(torch.from_numpy(batches[2]), wte=wte_d, wpe=wpe_d, exchange="gather_rows"), outs[2]))
        # a ticket that is never finished (the caller gave up on the batch) must not block the slot for ever
        lost = sh.gather_rows_begin(torch.from_numpy(batches[1]))
        sh.gather_rows_abandon(lost)
        after = bool(torch.equal(sh.embed_tokens(torch.from_numpy(batches[3]), wte=wte_d, wpe=wpe_d, exchange="gather_rows", check=True), outs[3]))
        q.put((rank, same and again and after, "", tuple(outs[0].shape), None))
        dist.barrier()
        sh.close()
        dist.destroy_process_group()
    except Exception as e:
        import traceback
        q.put((rank, False, repr(e) + traceback.format_exc(), None, None))


@pytest.mark.parametrize("fmt,d,max_n,world,head,chunks,slots,transport", [("int4", 1024, 3, 3, 100, 1, 2, "p2p"), ("int8", 768, 4, 2, 0, 3, 2, "p2p"),
                                                                           ("int4", 1024, 3, 2, 100, 1, 3, "p2p"), ("fp16", 768, 3, 3, 0, 2, 4, "p2p"),
                                                                           ("int4", 1024, 3, 3, 100, 1, 2, "sdma"), ("int8", 768, 3, 2, 50, 1, 3, "sdma")])
def test_split_phase_gather_two_batches_in_flight_across_processes(fmt, d, max_n, world, head, chunks, slots, transport):
    """gather_rows_begin / gather_rows_finish as a serving loop issues them: batch i + 1 is planned, packed and gathered on
    the cache's side stream (plan slot i + 1 mod 2) while batch i is reduced on the main stream.  Five different batches,
    real kernels, separate processes (gloo over one GPU): every output is bit-identical to the unsharded lookup of ITS batch."""
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_split_phase, args=(r, world, port, fmt, d, max_n, head, chunks, q, slots, transport)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        if p.is_alive():                                # (a rank left waiting for a peer that failed)
            p.terminate()
    for rank, same, err, shape, _ in results:
        assert shape is not None, f"rank {rank} failed: {err}"
        assert shape == (5, 33, d) and same, rank


def _worker_soak(rank, world, port, slots, q):
    try:
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        import torch.distributed as dist
        dist.init_process_group("gloo", rank=rank, world_size=world)
        torch.cuda.set_device(0)
        from scone_amd import EmbeddingCache, NGramExtractor
        from scone_amd.distributed import ShardedEmbeddingCache
        fmt, d, max_n, head = "int4", 1024, 3, 60
        keys, lens, table, _, wte, wpe = _problem(fmt, d, max_n)
        ex = NGramExtractor.from_arrays(keys, lens, max_n=max_n)
        sh = ShardedEmbeddingCache(ex, d, table_format=fmt, rank=rank, world=world, replicated_rows=head, plan_slots=slots)
        sh.load_rows(torch.from_numpy(table), 0)
        full = EmbeddingCache(ex, d, table_format=fmt)
        full.cache_embeddings(list(range(len(lens))), torch.from_numpy(table), verbose=False)
        rng = np.random.default_rng(int(os.environ.get("SCONE_SOAK_SEED", "2026")))   # the same choices on every rank
        wte_d = torch.from_numpy(wte).half().cuda()
        wpe_d = torch.from_numpy(np.tile(wpe, (3, 1))[:80]).half().cuda()
        n, bad, tickets, batches = int(os.environ.get("SCONE_SOAK_STEPS", "60")), [], [], []
        free0 = None
        transports = ["p2p", "all_gather"]
        if world == 2:                                                 # a HIP interprocess event survives 32 records: each slot keeps a
            sh._sdma_event_records, sh._sdma_event_pool = 3, 2         # pool of them and replaces it when used up -- here every 6 uses
        if sh.set_gather_transport("sdma") == "sdma":                 # collective: the capability probe + handle exchange, once
            transports.append("sdma")                                  # (round 4: the copy-engine transport joins the mix)
        used, hist = set(), []
        for i in range(n + slots - 1):
            if i < n:
                B, T = int(rng.integers(1, 12)), int(rng.integers(1, 80))
                tok = torch.from_numpy(rng.integers(0, 24, size=(B, T)))
                sh.gather_chunks = int(rng.integers(1, 4))             # 1 = columns on the wire, 2 / 3 = chunked records
                sh.shard_match = bool(rng.integers(2))                 # the plan's match sharded over the ranks, or not
                sh.gather_transport = transports[int(rng.integers(len(transports)))]
                used.add(sh.gather_transport if sh.gather_chunks == 1 else "chunked")
                hist.append((i, B, T, sh.gather_chunks, sh.shard_match, sh.gather_transport))
                tickets.append(sh.gather_rows_begin(tok))
                batches.append(tok)
            if i >= slots - 1:
                k = i - (slots - 1)
                out = sh.gather_rows_finish(tickets[k], wte=wte_d, wpe=wpe_d)
                if not torch.equal(out, full.embed_tokens(batches[k], wte=wte_d, wpe=wpe_d)):
                    bad.append(k)
                tickets[k] = None
            if i == 20:
                torch.cuda.synchronize()
                free0 = torch.cuda.mem_get_info()[0]
        torch.cuda.synchronize()
        drift = free0 - torch.cuda.mem_get_info()[0]
        if len(transports) == 3 and n >= 60 and "sdma" not in used:
            bad.append("the sdma transport was never drawn")
        q.put((rank, bad, int(drift), sh.table.status()))
        dist.barrier()
        sh.close()
        dist.destroy_process_group()
    except Exception as e:
        import traceback
        q.put((rank, repr(e) + traceback.format_exc() + f" last batches (i, B, T, chunks, shard_match, transport): {locals().get('hist', [])[-6:]}", 0, -1))


@pytest.mark.parametrize("world,slots", [(2, 2), (3, 3)])
def test_split_phase_soak_random_shapes_forms_and_slots(world, slots):
    """60 batches of random shape through the split-phase loop, `slots` batches in flight, every batch with its own form --
    one piece with columns on the wire or 2-3 chunks of records, match sharded over the ranks or not, exact ranges, padded
    all-gathers or copy-engine pushes into peer-mapped buffers (sdma): buffers of every slot are re-used and re-grown across shapes; every output equals the unsharded lookup of
    ITS batch, the status word stays clean, device memory does not drift."""
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_soak, args=(r, world, port, slots, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=1100) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        if p.is_alive():                                # (a rank left waiting for a peer that failed)
            p.terminate()
    for rank, bad, drift, status in results:
        assert isinstance(bad, list), f"rank {rank} failed: {bad}"
        assert bad == [] and status == 0, (rank, bad, status)
        assert drift < 256 * 2**20, (rank, drift)                       # (three processes share the card: allow their allocators some room)


@pytest.mark.parametrize("mode", ["replicated", "sharded", "sharded-slices"])
def test_bench_two_ranks_launched_like_the_driver(mode):
    """`python -m torch.distributed.run --nproc-per-node 2 bench.py --gpus 2` with the rehearsal knobs
    (gloo, both ranks on device 0): one JSON line, n_gpus 2, value = tokens of BOTH ranks / max time."""
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    env = dict(os.environ, SCONE_DIST_BACKEND="gloo", SCONE_ONE_DEVICE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"),
           "--gpus", "2", "--steps", "5", "--warmup", "2", "--rows", "200000", "--batch", "256", "--table-mode", mode.split("-")[0],
           "--sharded-rows-per-rank", "300000", "--sharded-steps", "2", "--pinned-rows", "300000", "--cpu-seconds", "1"]
    if mode == "sharded-slices":
        cmd.append("--no-gather-output")
    p = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    r = json.loads(lines[0])
    assert r["n_gpus"] == 2 and r["steps"] == 5 and "incomplete" not in r
    ranks_counted = 2 if mode == "replicated" else 1      # sharded: every rank embeds the SAME batch (strong scaling)
    assert r["scaling"] == ("weak" if mode == "replicated" else "strong")
    assert abs(r["value"] - ranks_counted * 256 * 512 * 5 / (r["ms_per_step"] * 5e-3)) / r["value"] < 1e-6
    if mode == "replicated":
        assert r["roofline"]["kernel_ms"]["n"] == 5
        km = r["roofline"]["kernel_ms"]
        assert km["n"] == 5 and km["min"] <= km["median"] <= km["max"]
        assert 0 < r["roofline"]["frac"] <= 1.0 and r["roofline"]["algorithmic_frac"] > 0
        cb = r["cpu_baseline"]                                   # N > 1: the short 1-core sample + the 8-sequence check
        assert cb["value"] > 0 and cb["cores"] == 1 and "python_all_cores" not in cb and cb["gpu_vs_oracle_max_rel_err"] < 1e-3
        assert r["world_sanity"]["all_gather_1KB_per_rank_ok"] is True and r["world_sanity"]["world_size"] == 2     # before anything else
        ex = r["sharded"]["exchanges"]                           # the printed line: one short entry per exchange ...
        assert all(e["ms_per_step"] > 0 and e["status_bits"] == 0 and e["speedup_vs_n1_pinned_host"] > 0 for e in ex.values()) and len(ex) == 6
        assert r["sharded"]["exchanges_agree"] is True
        _check_sharded_record(_details(r)["sharded"], 2)       # ... the whole record in the details file
    else:
        assert "sharded" not in r and "cpu_baseline" not in r


def _details(r):
    """The whole record behind a printed line (round 5: the line is the compact form, <= 6000 characters; `details` names the
    file with every phase split and provenance string)."""
    assert len(json.dumps(r)) <= 6500, len(json.dumps(r))
    path = r["details"] if os.path.isabs(r["details"]) else os.path.join(ROOT, r["details"])
    full = json.load(open(path))
    assert full["n_gpus"] == r["n_gpus"] and full["value"] == r["value"] and full["ms_per_step"] == r["ms_per_step"]
    return full


def _check_sharded_record(rec, world):
    """The C5-shaped sub-record of an N > 1 line (round 4: six exchanges behind a world sanity check): every exchange ran on
    `world` ranks -- phase split, wire bytes, a roofline block with physical fractions, the speed-up over the N = 1 baseline
    measured by rank 0 in the same process, the status bits, what the form costs a rank at other world sizes -- and they
    produced the same output."""
    assert "error" not in rec, rec
    assert rec["world_sanity"]["all_gather_1KB_per_rank_ok"] is True and rec["world_sanity"]["world_size"] == world
    assert rec["world_size"] == world and rec["device_count"] >= 1 and "n1_baseline" in rec
    assert rec["n1_pinned_host"]["value"] > 0 and rec["rows_total"] == world * rec["rows_per_rank"]
    assert "N = 2 / 4 / 8 ranks hold" in rec["workload"]
    one_call = ("rows_slices_only", "rows+all_gather", "gather_rows")
    split = ("gather_rows_split_phase", "gather_rows_split_phase_p2p", "gather_rows_split_phase_sdma")
    # plainest collective first, the newest transport last
    assert list(rec["exchanges"]) == [split[0], split[1]] + list(one_call) + [split[2]]
    for name in one_call:
        e = rec["exchanges"][name]
        assert "error" not in e, e
        assert e["ms_per_step"] > 0 and e["tokens_per_s"] > 0 and e["wire_bytes_received_rank0"] > 0 and e["status_bits"] == 0
        for ph in ("plan_ms", "pack_ms", "collective_ms", "embed_ms"):
            assert e["phase_ms_slowest_rank"][ph] >= 0.0, (name, ph)
        assert abs(e["speedup_vs_n1_pinned_host"] - e["tokens_per_s"] / rec["n1_pinned_host"]["value"]) < 1e-9
        rf = e["roofline"]
        assert 0 < rf["frac"] <= 1.0 and rf["per_rank_compulsory_bytes"] <= rf["per_rank_algorithmic_bytes"]
        assert rf["wire"]["bytes_received_rank0"] == e["wire_bytes_received_rank0"] and rf["wire"]["GBps"] > 0
    assert rec["exchanges"]["rows_slices_only"]["roofline"]["per_rank_tokens_reduced"] < \
        rec["exchanges"]["gather_rows"]["roofline"]["per_rank_tokens_reduced"]
    assert "gather_out_ms" in rec["exchanges"]["rows+all_gather"]["phase_ms_slowest_rank"]
    # which forms scale: only the slice exchange divides a rank's HBM bytes by the world size
    sl, gr = rec["exchanges"]["rows_slices_only"], rec["exchanges"]["gather_rows"]
    assert sl["scales_with_world"] is True and gr["scales_with_world"] is False and rec["exchanges"]["rows+all_gather"]["scales_with_world"] is False
    assert sl["per_rank_hbm_bytes_vs_world"]["8"] < 0.3 * sl["per_rank_hbm_bytes_vs_world"]["2"]
    assert gr["per_rank_hbm_bytes_vs_world"]["8"] == gr["per_rank_hbm_bytes_vs_world"]["2"]
    assert "rows_slices_only" in rec["form_for_data_parallel_consumers"]
    assert rec["exchanges_agree"] is True and len(rec["exchanges_compared"]) == 5 and not any(rec["status_bits"].values())
    for name in split:
        sp = rec["exchanges"][name]
        assert "error" not in sp, sp
        assert sp["ms_per_step"] > 0 and sp["batches_in_flight"] == 3 and sp["status_bits"] == 0
        assert sp["speedup_vs_n1_pinned_host"] > 0 and 0 < sp["roofline"]["frac"] <= 1.0 and sp["scales_with_world"] is False
        if name.endswith("sdma"):
            assert sp["with_cu_reserve"] is None                 # no transport kernel to make room for
        else:                                                    # the same loop on the CU-masked stream
            assert sp["with_cu_reserve"]["compute_units_reserved"] == 32 and sp["with_cu_reserve"]["ms_per_step"] > 0
        assert sp["sync_free_plan"]["exchanges"] >= 1            # from the second batch on nothing waits for the device (all three transports)
    # the copy-engine transport really ran (interprocess handles work between processes on one device as well)
    sd = rec["exchanges"]["gather_rows_split_phase_sdma"]
    assert sd["transport_fallback_reason"] is None and sd["records_transport"].startswith("copy-engine"), sd
    best = rec["best_whole_output"]
    assert best["exchange"] != "rows_slices_only" and best["tokens_per_s"] == rec["exchanges"][best["exchange"]]["tokens_per_s"]


def test_bench_starts_its_own_ranks_when_no_launcher_did():
    """`python bench.py --gpus 2` with WORLD_SIZE unset: bench.py spawns the two ranks itself (before it touches a GPU) and
    forwards ONE line that says n_gpus = 2; `--gpus 64` on this box exits non-zero instead of printing a 1-GPU line."""
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update(SCONE_DIST_BACKEND="gloo", SCONE_ONE_DEVICE="1", HSA_ENABLE_IPC_MODE_LEGACY="0", PYTHONFAULTHANDLER="1")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--rows", "200000",
           "--batch", "128", "--sharded-rows-per-rank", "200000", "--sharded-steps", "1", "--pinned-rows", "200000",
           "--cpu-seconds", "1"]
    p = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    r = json.loads(lines[0])
    assert r["n_gpus"] == 2 and "started 2 ranks itself" in r["launcher"]
    _check_sharded_record(_details(r)["sharded"], 2)
    env.pop("SCONE_ONE_DEVICE")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "64", "--steps", "1"], env=env, cwd=ROOT,
                       capture_output=True, text=True, timeout=120)
    assert p.returncode != 0 and not [ln for ln in p.stdout.splitlines() if ln.startswith("{")]


def _rccl_world1_worker(port, q):
    try:
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0", SCONE_DIST_TRACE="1")
        import torch.distributed as dist
        torch.cuda.set_device(0)
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
        from scone_amd import distributed as D
        dev = "cuda"
        done = []
        # the dtypes and shapes the sharded path hands to RCCL (one rank: every collective is a copy to itself, but argument
        # validation, dtype support and stream semantics are the real thing)
        ell_send = torch.arange(64 * 8, dtype=torch.int32, device=dev).view(64, 8)
        ell = torch.empty_like(ell_send)
        D._all_gather(ell.view(-1), ell_send.view(-1), None)                         # list records: int32
        assert torch.equal(ell, ell_send); done.append("all_gather int32")
        cnt = torch.tensor([12345], dtype=torch.int64, device=dev)
        allc = torch.empty(1, dtype=torch.int64, device=dev)
        D._all_gather(allc, cnt, None)                                              # record counts: int64 -> .tolist()
        assert allc.tolist() == [12345]; done.append("all_gather int64")
        rows = torch.randint(0, 255, (33, 512), dtype=torch.uint8, device=dev)
        got = torch.empty_like(rows)
        D._all_gather_async(got.view(-1), rows.reshape(-1), None).wait()            # a column, padded all-gather form: uint8, async
        assert torch.equal(got, rows); done.append("all_gather uint8 async")
        frag = torch.arange(128, dtype=torch.int64, device=dev)
        gf = torch.empty_like(frag)
        D._all_gather_async(gf, frag, None).wait()                                  # hash fragments: int64
        assert torch.equal(gf, frag); done.append("all_gather int64 async")
        recs = torch.randint(0, 255, (57, 544), dtype=torch.uint8, device=dev)
        out = torch.empty_like(recs)
        D._all_to_all(out, recs, [57], [57], None)                                  # slice exchange: uint8 [n, record], uneven splits
        assert torch.equal(out, recs); done.append("all_to_all_single uint8 2-D with splits")
        part = torch.randn(40, 1024, device=dev)
        mine = torch.empty_like(part)
        D._reduce_scatter_sum(mine, part, None)                                     # partial sums: fp32
        assert torch.equal(mine, part); done.append("reduce_scatter_tensor fp32")
        w = D._exchange_exact_async(rows, [0, 33], [33], 0, None)                   # exact ranges with no peer: nothing to send
        w.wait(); done.append("batch_isend_irecv (no peers)")
        t64 = torch.tensor([1.5], dtype=torch.float64, device=dev)                  # bench.py's timing reductions
        dist.all_reduce(t64, op=dist.ReduceOp.MAX)
        dist.all_reduce(t64, op=dist.ReduceOp.MIN)
        assert float(t64.item()) == 1.5; done.append("all_reduce float64 MAX / MIN")
        objs = [None]                                                               # the sdma transport's host side: handles travel as
        dist.all_gather_object(objs, (b"\0" * 64, [b"\1" * 64] * 4, None))          # pickled objects over the group ...
        assert objs[0][1][3] == b"\1" * 64
        ctrl = dist.new_group(ranks=[0], backend="gloo")                            # ... and its rendezvous on a gloo group beside it
        dist.barrier(group=ctrl); done.append("all_gather_object + a gloo group beside the RCCL one")
        dist.barrier()
        torch.cuda.synchronize()
        q.put((done, None))
        dist.destroy_process_group()
    except Exception as e:
        import traceback
        q.put((None, repr(e) + traceback.format_exc()))


def test_rccl_accepts_every_collective_of_the_sharded_path_world1():
    """RCCL itself (backend "nccl"), ONE rank on the one GPU: every collective the sharded path and bench.py issue -- with the
    dtypes, shapes, split lists and async handles they use -- is accepted, completes and delivers (to itself).  Not a
    substitute for the multi-GPU test below, but argument validation, dtype support (uint8 / int32 / int64 / float64), the
    dmabuf-IPC environment and the stream semantics are RCCL's own here, on every box of this pool."""
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_rccl_world1_worker, args=(_free_port(), q))
    p.start()
    done, err = q.get(timeout=300)
    p.join(timeout=60)
    assert done is not None, err
    assert len(done) == 9, done


def test_bench_runs_under_rccl_with_one_rank():
    """`bench.py --force-dist` on one GPU: the N > 1 code path of the headline (process group with `device_id`, barrier +
    synchronise around the timed region, max-over-ranks all-reduce on the device, final barrier, destroy) under RCCL."""
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "SCONE_DIST_BACKEND", "SCONE_ONE_DEVICE")}
    env.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), HSA_ENABLE_IPC_MODE_LEGACY="0")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--force-dist", "--quick", "--steps", "5", "--warmup", "2",
                        "--rows", "200000", "--batch", "256"], env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith('{"metric"')]
    assert len(lines) == 1, p.stdout[-2000:]
    r = json.loads(lines[0])
    assert r["n_gpus"] == 1 and r["value"] > 0 and r["roofline"]["kernel_ms"]["n"] == 5


def _nccl_worker(rank, world, port, q):
    try:
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        import torch.distributed as dist
        torch.cuda.set_device(rank)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", rank))
        from scone_amd import EmbeddingCache, NGramExtractor
        from scone_amd.distributed import ShardedEmbeddingCache
        res = []
        for fmt, d, max_n, head, chunks in (("int8", 768, 3, 0, 1), ("int4", 1024, 3, 100, 3), ("fp16", 768, 4, 37, 4)):
            keys, lens, table, tok, wte, wpe = _problem(fmt, d, max_n)
            ex = NGramExtractor.from_arrays(keys, lens, max_n=max_n)
            sh = ShardedEmbeddingCache(ex, d, table_format=fmt, rank=rank, world=world, replicated_rows=head,
                                       gather_chunks=chunks)
            sh.load_rows(torch.from_numpy(table), 0)
            wte_d, wpe_d = torch.from_numpy(wte).half().cuda(), torch.from_numpy(wpe).half().cuda()
            full = EmbeddingCache(ex, d, table_format=fmt)
            full.cache_embeddings(list(range(len(lens))), torch.from_numpy(table), verbose=False)
            ref = full.embed_tokens(torch.from_numpy(tok), wte=wte_d, wpe=wpe_d)
            for exchange in ("rows", "gather_rows", "gather_rows:all_gather", "partial_sums",
                             "rows::sm", "gather_rows::sm", "gather_rows:all_gather:sm", "gather_rows::sm:c1", "gather_rows:all_gather::c1"):
                sh.gather_chunks = 1 if exchange.endswith(":c1") else chunks     # one piece: columns on the wire
                sh.gather_transport = exchange.split(":")[1] if ":" in exchange and exchange.split(":")[1] else "p2p"   # exact p2p ranges / padded all-gather
                sh.shard_match = ":sm" in exchange                               # the plan's match sharded over the ranks (+ one all-gather)
                got = sh.embed_tokens(torch.from_numpy(tok), wte=wte_d, wpe=wpe_d, exchange=exchange.partition(":")[0])
                err = float((got.float() - ref.float()).abs().max() / ref.float().abs().max())
                res.append((fmt, exchange, bool(torch.equal(got, ref)), err))
            sh.gather_transport, sh.shard_match, sh.gather_chunks = "p2p", True, chunks
            # the split-phase loop: two batches in flight, the second one's transfers behind the first one's reduction
            tk = sh.gather_rows_begin(torch.from_numpy(tok))
            for i in range(3):
                got = sh.gather_rows_finish(tk, wte=wte_d, wpe=wpe_d)
                tk = sh.gather_rows_begin(torch.from_numpy(tok)) if i < 2 else None
                res.append((fmt, f"gather_rows split-phase step {i}", bool(torch.equal(got, ref)), 0.0))
            # the copy-engine transport ACROSS devices (the one thing the one-GPU tests cannot show): self-test at set-up, then the
            # same loop; a fallback (reason kept in the result) still has to give the right rows
            sh.gather_chunks = 1
            used = sh.set_gather_transport("sdma")
            tk = sh.gather_rows_begin(torch.from_numpy(tok))
            for i in range(40):                                                  # past the 32 records of one interprocess event
                got = sh.gather_rows_finish(tk, wte=wte_d, wpe=wpe_d)
                tk = sh.gather_rows_begin(torch.from_numpy(tok)) if i < 39 else None
                if i in (0, 1, 39) or not torch.equal(got, ref):
                    res.append((fmt, f"gather_rows sdma ({used}; {sh.transport_fallback_reason}) step {i}", bool(torch.equal(got, ref)), 0.0))
            dist.barrier()
            sh.close()
        q.put((rank, res, None))
        dist.barrier()
        dist.destroy_process_group()
    except Exception as e:
        import traceback
        q.put((rank, None, repr(e) + traceback.format_exc()))


def test_sharded_exchanges_under_rccl_one_rank_per_gpu():
    """The first box with two or more GPUs runs this: backend "nccl" (= RCCL), one rank per device, all three exchanges
    -- all_to_all_single with uneven uint8 splits, the chunked all_gather_into_tensor of records, reduce_scatter_tensor
    of fp32 sums, and the all-gather of the finished vectors -- on device buffers over xGMI, checked against the
    unsharded table on the same GPUs (row exchanges bit-identical).  Skipped on one-GPU boxes."""
    world = min(torch.cuda.device_count(), 6)           # the pool allows at most 6 processes on the GPUs at once
    if world < 2:
        pytest.skip("needs two or more GPUs (RCCL refuses two ranks on one device)")
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_nccl_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=600) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        if p.is_alive():                                # (a rank left waiting for a peer that failed)
            p.terminate()
    for rank, res, err in results:
        assert res is not None, f"rank {rank} failed: {err}"
        for fmt, exchange, same, e in res:
            if exchange.startswith("partial_sums"):
                assert e < 1e-3, (rank, fmt, exchange, e)
            else:
                assert same, (rank, fmt, exchange, e)
