#!/usr/bin/env python3
"""bench.py -- f-gram embed throughput on MI355X (BASELINE.json's metric).

    python bench.py [--gpus N] [--steps K] [--warmup W]

One "step" = one pass of the hot path (n-gram match -> INT8 row gather -> dequantise ->
mean -> + wte + wpe -> fp16 store; `scone_embed`) over one batch of B x T synthetic tokens
already resident in HBM.  Headline workload (N = 1 and every rank at N > 1): 1M-row INT8
f-gram table, d = 768, max_n = 3, GPT-2 vocabulary, S_uniform stream.

N > 1: the 1M-row table fits one GPU, sequences are independent, so the path shards over
tokens -- every rank holds the table and embeds its own batch; no data-path collective;
"scaling": "weak".  Launched either by the driver's `python -m torch.distributed.run ... bench.py
--gpus N` (RANK / LOCAL_RANK / WORLD_SIZE in the environment) or as plain `python bench.py --gpus N`,
which starts the N ranks itself (fresh child processes, before this process touches a GPU) and
fails unless the line it forwards says n_gpus == N.

Prints ONE JSON line on rank 0 (contract in the task statement), at most 6000 characters (`compact_record`: the driver keeps
the tail of stdout); the whole record -- every workload description, provenance string and phase split -- is written to the file
the line names (`details`: gpurun_out/bench_details_n<N>.json, or $SCONE_BENCH_DETAILS).  The record holds
  roofline      the gather/reduce kernel, HIP-event timed on its launch stream.  `frac` is a PHYSICAL fraction of the
                8 TB/s HBM peak, never above 1: the bytes that left L2 (committed rocprofv3 PMC passes, quoted only while
                the kernel's sources are the ones they were taken on) / kernel time / peak -- or, without such an entry,
                the compulsory bytes (every distinct row once + output + ids).  SURVEY 8d's algorithmic figure (every
                row REFERENCE counted; cache reuse can carry it past the peak) is `algorithmic_frac`.  Also: min /
                median / max launch time, `match_us`, and the same figures for a cache-defeating variant (`hbm_variant`)
  cpu_baseline  the line-for-line Python port of the reference loop (oracle/ref_port.py), 1 core; at N = 1 beside it the
                same port on all host cores (multiprocessing over sequences) and the plain-C oracle with OpenMP; the GPU
                output of 8 sequences drawn from the whole batch is checked against it
  sharded       N > 1: the row-sharded path on the C5-shaped workload (INT4 d = 1024, 125M rows per rank, replicated
                index, 1M-token batch) for every exchange, each with phase split, wire bytes, a roofline block (per-rank
                HBM bytes / step / 8 TB/s; wire bytes / collective time / xGMI peak) and `speedup_vs_n1_pinned_host`:
                rank 0 measures the single-GPU alternative (pinned host DRAM, C4-shaped) in the same process, so the
                north-star's ">= 4x at 8 GPUs vs 1 GPU" is readable from ONE record.  N = 1: that baseline alone, on the
                S_uniform stream (rows read in place over PCIe) and on a Zipf stream (staged, de-duplicated prefetch)

Time budget.  `--time-budget S` (default 380 s, inside the driver's 600 s) is ABSOLUTE, from the start of the first
bench.py process of the job (taken before `import torch`, which can cost a minute or two on a cold box).  A watchdog thread on every rank enforces it and the per-stage limits inside it (every collective
of the `sharded` record runs in a stage): on expiry rank 0 prints the line with everything measured so far -- marked
`incomplete` -- and every rank leaves through os._exit (a hung RCCL collective cannot be cancelled; no process that
touched the GPU is ever replaced by another program).  The exit status is 0 when the headline metric was measured
(the line is valid; the record says what is missing) and 3 when it was not.
"""

import time
_T_PROCESS_START = time.time()      # before the heavy imports: a cold `import torch` can take a minute or two of the budget

import argparse
import hashlib
import json
import os
import re
import socket
import subprocess
import sys
import threading

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
# the host driver only supports dmabuf IPC: must be in the environment BEFORE the HIP runtime starts (RCCL at N > 1)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

from benchkit.common import HBM_PEAK_GBPS, PCIE_PEAK_GBPS, T0_ENV, XGMI_LINK_GBPS_PER_DIRECTION  # noqa: E402,F401
from benchkit.record import LINE_LIMIT, Budget, Line, Watchdog, compact_record, details_path_for  # noqa: E402,F401
from benchkit.launcher import self_launch  # noqa: E402
from benchkit.roofline import (_code_only, kernel_source_files, kernel_source_sha, kernel_stats, read_traffic,  # noqa: E402,F401
                               roofline_block, workload_bytes, workload_sig)
from benchkit.workloads import (MAX_DISTINCT_BATCHES, REFERENCE_GRID, c1_record, config_record, hbm_variant, latency_block, lookup_loop,  # noqa: E402,F401
                                make_batches, make_vocabulary, measure_lookup, pinned_baseline)
from benchkit.sharded import NCCL_HIGH_PRIORITY, _rccl_version, run_stages, sharded_record  # noqa: E402,F401


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--rows", type=int, default=1_000_000)
    ap.add_argument("--dim", type=int, default=768)
    ap.add_argument("--format", default="int8", choices=["fp32", "fp16", "int8", "int4"])
    ap.add_argument("--batch", type=int, default=2048)
    ap.add_argument("--seq", type=int, default=512)
    ap.add_argument("--stream", default="uniform", choices=["uniform", "zipf"])
    ap.add_argument("--cache-rows", type=int, default=0, help="--placement pinned_host with --stage-tokens: row slots of the HBM cache of "
                    "cold rows (0 = the pipeline's minimum)")
    ap.add_argument("--sharded-cu-reserve", type=int, default=32, help="sharded record: the split-phase stages over RCCL transports "
                    "are timed a second time with this many compute units left to the transport kernels (0 = skip)")
    ap.add_argument("--pinned-cache-rows", type=int, default=16_000_000, help="n1_pinned_host_zipf: row slots of the HBM cache of cold rows")
    ap.add_argument("--pinned-stage-tokens", type=int, default=262144, help="n1_pinned_host_zipf: tokens per chunk of the prefetch pipeline")
    ap.add_argument("--pinned-zipf-steps", type=int, default=20)
    ap.add_argument("--pinned-zipf-warmup", type=int, default=400, help="n1_pinned_host_zipf: batches that warm the cache before the timed steps")
    ap.add_argument("--keygen", default="zipf", choices=["zipf", "zipf_gpu", "structured"],
                    help="vocabulary generator: seeded Zipf n-grams with de-duplication on the host (default), the same law "
                         "drawn and de-duplicated on the GPU (seconds instead of minutes at 1e7 rows: config C3), or the "
                         "distinct-by-construction generator for >= 1e8 rows")
    ap.add_argument("--vocab", type=int, default=50257, help="token vocabulary (structured keygen only; <= 262144, the direct unigram "
                    "table's size).  262144 with --rows 10000000: the token-indexed rows of a launch (wte + unigram rows, 0.6 GB) no "
                    "longer fit the 256-MB Infinity Cache -- the variant behind roofline.mall_variant")
    ap.add_argument("--same-batch", action="store_true", help="re-use ONE batch for every step (rounds 1-4; the Infinity Cache "
                    "then carries rows from step to step).  Default: a different batch every step")
    ap.add_argument("--prefetch", default="auto", choices=["auto", "on", "off"],
                    help="scone_embed_prefetch of batch i + 1 right after the lookup of batch i (a pinned-host table with "
                         "--stage-tokens: the next batch's first chunks are matched, placed and copied beside this batch's last "
                         "lookups; a no-op for every other table).  auto = on where it does something")
    ap.add_argument("--no-configs", action="store_true", help="skip the `configs` block (BASELINE configs C1, C2, C3, C4-in-HBM)")
    ap.add_argument("--out-candidates", type=int, default=8, help="the re-used [B, T, d] output buffer is the fastest of this many "
                    "candidate allocations (5 timed lookups each, before the timed region); 1 = one plain allocation")
    ap.add_argument("--no-latency", action="store_true", help="skip the `latency` block (the reference's benchmark grid on the headline table)")
    ap.add_argument("--configs-steps", type=int, default=15)
    ap.add_argument("--placement", default="hbm", choices=["hbm", "pinned_host"])
    ap.add_argument("--hot-rows", type=int, default=0, help="pinned_host: leading rows kept in HBM")
    ap.add_argument("--stage-tokens", type=int, default=0, help="pinned_host: staged prefetch chunk size (0 = zero-copy)")
    ap.add_argument("--table-mode", default="replicated", choices=["replicated", "sharded"])
    ap.add_argument("--exchange", default="auto", choices=["auto", "rows", "gather_rows", "partial_sums"],
                    help="sharded mode: all-to-all of quantised rows + all-gather of the output (default), all-gather of the "
                         "quantised rows with every rank reducing the whole batch, or reduce-scatter of fp32 partial sums")
    ap.add_argument("--replicated-rows", type=int, default=50257,
                    help="sharded mode: head of the table kept on every rank (default: the unigram rows)")
    ap.add_argument("--no-gather-output", action="store_true",
                    help="sharded mode: stop after every rank has finished its own slice of the batch (a consumer that is "
                         "data-parallel over the same slices needs no all-gather of the [B, T, d] output)")
    ap.add_argument("--shard-of", default="", help="R/W: build only shard R of a W-way row-sharded table on this one GPU "
                    "and time its local work (partial sums + finalise of its 1/W token slice); no exchange -- "
                    "capacity / kernel check for tables that need W GPUs (C5)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-hbm-variant", action="store_true", help="skip the cache-defeating variant behind roofline.hbm_variant")
    ap.add_argument("--no-sharded-record", action="store_true",
                    help="skip the `sharded` sub-record (N > 1: C5-shaped row-sharded run; N = 1: pinned-host baseline)")
    ap.add_argument("--quick", action="store_true", help="headline measurement only (A/B tools): no cpu baseline, no "
                    "hbm variant, no sharded record")
    ap.add_argument("--sharded-rows-per-rank", type=int, default=125_000_000,
                    help="sharded record: table rows per rank (C5: 1e9 rows over 8 GPUs)")
    ap.add_argument("--sharded-steps", type=int, default=5)
    ap.add_argument("--pinned-rows", type=int, default=100_000_000, help="rows of the pinned-host table (C4) behind "
                    "sharded.n1_pinned_host: the single-GPU baseline of the row-sharded record")
    ap.add_argument("--force-dist", action="store_true", help="init torch.distributed even with one rank (tests the N>1 code path)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--time-budget", type=float, default=380.0,
                    help="absolute limit in seconds from the start of the job's first bench.py process (driver timeout: 600; a "
                         "launcher's own cold start comes on top): when it is used up rank 0 prints the line with what has "
                         "been measured and every rank exits")
    ap.add_argument("--stage-limit", type=float, default=90.0,
                    help="limit in seconds for one stage of the sharded record (one exchange with its collectives)")
    ap.add_argument("--selftest", default="", choices=["", "hang", "ok"], help=argparse.SUPPRESS)   # CPU rehearsal of the watchdog
    a = ap.parse_args(argv)
    if a.quick:
        a.no_cpu_baseline = a.no_hbm_variant = a.no_sharded_record = a.no_configs = a.no_latency = True
    return a


# ----------------------------------------------------------------------------------------------------------------
# the cpu_baseline leg: the ONLY code outside tests/ and smoke() that imports oracle/ (the checker and the timed CPU baseline)
def _cpu_pool_init(keys, lens, ids, rows, d):
    global _POOL_CACHE
    import torch
    from oracle import ref_port as R
    torch.set_num_threads(1)
    f2id = {}
    for k, l, i in zip(keys.tolist(), lens.tolist(), ids.tolist()):
        f2id[tuple(k[:l])] = i
    c = R.RefCache(f2id, 3, d)
    for i, r in zip(ids.tolist(), rows):
        c.embeddings[i] = r
    _POOL_CACHE = (c, d)


def _cpu_pool_work(seqs):
    from oracle import ref_port as R
    c, d = _POOL_CACHE
    for s in seqs:
        R.aggregate(c, s, d)
    return len(seqs)


def cpu_baseline(args, keys, lens, tok, seed, base_scale, gpu_out, wte, wpe, seconds=None, all_cores=True):
    """Time the reference loop (set-of-tuples match -> dict id map -> torch.stack of fp32 rows
    -> mean -> zero-filled [1,T,d]; n_gram_extractor.py:106-126, embedding_cache.py:113-181,
    engine.py:234-266) on a bounded sample of the same stream, 1 core, and use its output to
    check the GPU result of 8 sequences drawn from the WHOLE batch (the last one always among them: every part of the
    large-batch kernel's walk -- first, middle and last iteration of a workgroup, the partial last block -- is looked at).
    `all_cores=False` (N > 1 lines): the 1-core figure and the check only."""
    import numpy as np
    import torch
    if keys is None or keys.shape[0] > 20_000_000:
        raise RuntimeError("cpu baseline skipped: a Python dict of > 2e7 f-grams (or of a structured vocabulary without host "
                           "key arrays) does not fit the time budget")
    from oracle import ref_port as R
    torch.set_num_threads(1)
    d = args.dim
    seconds = args.cpu_seconds if seconds is None else seconds
    f2id = R._key_dict(keys, lens)
    cache = R.RefCache(f2id, 3, d)

    def rows_for(ids):
        ids = np.asarray(sorted(ids), dtype=np.int64)
        if args.format == "int4":
            deq = R.dequantize_i4(*R.synth_rows_i4(seed, ids, d, base_scale))
        else:
            deq = R.synth_rows_i8(seed, ids, d).astype(np.float32) * \
                R.synth_scale_f16(seed, ids, base_scale).astype(np.float32)[:, None]
        if args.format == "fp16":
            deq = deq.astype(np.float16).astype(np.float32)
        return ids, deq

    def load_rows(seqs):
        need = set()
        for s in seqs:
            off, ids = R.match_csr_python(f2id, 3, s)
            need.update(int(i) for i in ids)
        need -= set(cache.embeddings.keys())
        if need:
            ids, deq = rows_for(need)
            for i, r in zip(ids.tolist(), deq):
                cache.embeddings[i] = r

    B = tok.shape[0]
    seqs = [tok[b].tolist() for b in range(min(B, 128 if all_cores else 32))]
    load_rows(seqs)                       # host copies of the rows the sample touches (not timed)
    # whole passes over the sample until ~seconds of CPU work have been timed
    done, dt = 0, 0.0
    t0 = time.perf_counter()
    while dt < seconds:
        for s in seqs:
            R.aggregate(cache, s, d)
        done += len(seqs)
        dt = time.perf_counter() - t0
    nseq = done
    py_all = c_line = None
    if all_cores:
        refs = np.asarray(sorted(cache.embeddings.keys()), dtype=np.int64)
        sub = np.stack([cache.embeddings[int(i)] for i in refs])
        # courtesy upper bound 1: the SAME Python port on all host cores, multiprocessing over
        # independent sequences (spawned workers -- this process holds a GPU context -- each with the sample's vocabulary)
        try:
            import multiprocessing as mp
            ncpu = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
            nw = max(1, min(ncpu, 16))
            ctx = mp.get_context("spawn")
            with ctx.Pool(nw, initializer=_cpu_pool_init, initargs=(keys[refs], lens[refs], refs, sub, d)) as pool:
                pool.map(_cpu_pool_work, [seqs[:1]] * nw)           # workers up and warm
                reps, t0 = 0, time.perf_counter()
                while time.perf_counter() - t0 < 5.0:
                    pool.map(_cpu_pool_work, [seqs[i::nw] for i in range(nw)])
                    reps += 1
                el = time.perf_counter() - t0
            py_all = {"value": reps * len(seqs) * tok.shape[1] / el, "unit": "tokens/s", "cores": nw,
                      "kind": "port (oracle/ref_port.py aggregate(), multiprocessing over sequences)",
                      "sample": f"{reps} passes over {len(seqs)} sequences x {tok.shape[1]} tokens ({el:.1f} s)"}
        except Exception as e:
            py_all = {"value": None, "error": repr(e)}
        # courtesy upper bound 2: the plain-C oracle (oracle/oracle.c), OpenMP over independent sequences on all host
        # cores, on the same sample (vocabulary and table restricted to the f-grams the sample references)
        try:
            from oracle.c_oracle import COracle
            co = COracle(keys[refs], lens[refs], 3)
            tok_s = np.asarray(seqs, dtype=np.int64)
            nthr = min(os.cpu_count() or 1, 64)
            co.embed(sub, tok_s[:8], "mean", nthr)
            reps, t0 = 0, time.perf_counter()
            while time.perf_counter() - t0 < 3.0:
                co.embed(sub, tok_s, "mean", nthr)
                reps += 1
            c_line = {"value": reps * tok_s.size / (time.perf_counter() - t0), "unit": "tokens/s", "cores": nthr,
                      "kind": "port (plain C, OpenMP over sequences; oracle/oracle.c)"}
        except Exception as e:
            c_line = {"value": None, "error": repr(e)}
    # parity check of the GPU output against the oracle: 8 sequences from all over the batch
    rng = np.random.default_rng(20260304)
    picks = sorted(set(int(x) for x in rng.choice(B, size=min(7, B), replace=False)) | {B - 1})
    check = [tok[b].tolist() for b in picks]
    load_rows(check)
    err = 0.0
    for b, s in zip(picks, check):
        fg = R.aggregate(cache, s, d)
        ref = R.combine(torch.from_numpy(tok[b:b + 1]), fg, wte.float().cpu(), wpe.float().cpu()).numpy()
        err = max(err, float(np.abs(gpu_out[b:b + 1].float().cpu().numpy() - ref).max() / np.abs(ref).max()))
    res = {
        "value": nseq * tok.shape[1] / dt, "unit": "tokens/s", "cores": 1, "kind": "port",
        "sample": f"{nseq} sequences x {tok.shape[1]} tokens ({len(seqs)} distinct sequences of the same stream, repeated) "
                  f"({dt:.1f} s; oracle/ref_port.py aggregate(), python {sys.version_info.major}.{sys.version_info.minor}, "
                  f"torch {torch.__version__}, host cpus {os.cpu_count()})",
        "gpu_vs_oracle_max_rel_err": err, "gpu_vs_oracle_sequences": picks,
    }
    if all_cores:
        res["python_all_cores"] = py_all
        res["c_oracle_all_cores"] = c_line
    return res


def cpu_baseline_spot_check(n_rows, keys, lens, tok_np, gpu_out, fmt, d, seed, base_scale, wte, wpe, n_pick=8, vocab=50257):
    """The checker half of the cpu_baseline leg for the `configs` block: the GPU output of `n_pick` sequences drawn from the
    whole batch (the last one always among them) against the numpy oracle (oracle/ref_port.py: match_hits -> hits_to_csr ->
    embed_numpy -> combine = n_gram_extractor.py:106-126, embedding_cache.py:113-181, engine.py:234-266,
    language_model.py:239-254) on the dequantised rows those sequences reference, recomputed on the host from the
    counter-based generator.  keys None: the structured vocabulary, matched through the closed-form inverse of its generator
    (match_hits_structured).  Returns (max relative error, the sequences)."""
    import numpy as np
    import torch
    from oracle import ref_port as R
    B = tok_np.shape[0]
    rng = np.random.default_rng(20260304)
    picks = sorted(set(int(x) for x in rng.choice(B, size=min(n_pick - 1, B), replace=False)) | {B - 1})
    sub = np.ascontiguousarray(tok_np[picks])
    hits = R.match_hits_structured(n_rows, sub, 3, vocab=vocab) if keys is None else R.match_hits(keys, lens, sub, 3)
    off, ids = R.hits_to_csr(hits)
    uniq = np.unique(ids)
    if fmt == "int4":
        rows = R.dequantize_i4(*R.synth_rows_i4(seed, uniq, d, base_scale))
    else:
        rows = R.synth_rows_i8(seed, uniq, d).astype(np.float32) * R.synth_scale_f16(seed, uniq, base_scale).astype(np.float32)[:, None]
        if fmt == "fp16":
            rows = rows.astype(np.float16).astype(np.float32)
    fg = R.embed_numpy(rows, off, np.searchsorted(uniq, ids), "mean").reshape(len(picks), tok_np.shape[1], d)
    ref = R.combine(torch.from_numpy(sub), torch.from_numpy(fg), wte.float().cpu(), wpe.float().cpu()).numpy()
    got = gpu_out[torch.tensor(picks, device=gpu_out.device)].float().cpu().numpy()
    return float(np.abs(got - ref).max() / np.abs(ref).max()), picks


def c1_oracle_leg(keys, lens, table, tok_np, gpu_out, seconds):
    """C1's half of the cpu_baseline leg: EVERY token of the batch against the numpy oracle (bit-exact: an fp32 table, the
    reference's own sequential mean), and the line-for-line Python port (RefCache + aggregate = embedding_cache.py:113-181,
    engine.py:234-266) timed on 1 core on the SAME 8 sequences."""
    import numpy as np
    import torch
    from oracle import ref_port as R
    torch.set_num_threads(1)
    ro, ri = R.hits_to_csr(R.match_hits(keys, lens, tok_np, 3))
    ref = R.embed_numpy(table, ro, ri, "mean").reshape(gpu_out.shape)
    exact = bool(np.array_equal(gpu_out, ref))
    err = float(np.abs(gpu_out - ref).max() / np.abs(ref).max())
    cache = R.RefCache(R._key_dict(keys, lens), 3, table.shape[1])
    for i in np.unique(ri).tolist():
        cache.embeddings[i] = table[i]
    seqs = [tok_np[b].tolist() for b in range(tok_np.shape[0])]
    done, dt, t0 = 0, 0.0, time.perf_counter()
    while dt < seconds:
        for sq in seqs:
            R.aggregate(cache, sq, table.shape[1])
        done += len(seqs)
        dt = time.perf_counter() - t0
    return {"gpu_vs_oracle_bit_exact": exact, "gpu_vs_oracle_max_rel_err": err, "gpu_vs_oracle_tokens": int(tok_np.size),
            "cpu_port_1core_tokens_per_s": done * tok_np.shape[1] / dt,
            "cpu_port_sample": f"{done} sequences x {tok_np.shape[1]} tokens ({dt:.1f} s; oracle/ref_port.py aggregate(), 1 core)"}


# ----------------------------------------------------------------------------------------------------------------
def selftest_main(args):
    """CPU rehearsal of the time budget (tests/test_bench_budget.py): no GPU, gloo, a made-up headline clearly marked as
    such, then two stages of `run_stages` -- the second one a collective that rank 1 never joins when `--selftest hang`.
    Everything that matters is the real code: Budget, Line, Watchdog, run_stages, self_launch, the exit path."""
    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    budget = Budget(args.time_budget, _T_PROCESS_START)
    line = Line(rank)
    watchdog = Watchdog(budget, line, rank)
    watchdog.start()
    time.sleep(0.5)                                # (a budget that is already gone ends the job here, before any "headline")
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo")
    res = {"metric": "selftest (no measurement: watchdog rehearsal on CPU)", "value": 0.0, "unit": "tokens/s", "n_gpus": world,
           "steps": args.steps, "warmup": args.warmup, "ms_per_step": 0.0, "higher_is_better": True, "scaling": "weak",
           "vs_baseline": None, "dtype": "none", "data": "selftest", "config": {"workload": "none"}, "selftest": args.selftest,
           "time_budget_s": budget.seconds}
    line.headline_done = True
    rec = {"exchanges": {}}
    if rank == 0:
        res["sharded"] = rec
    line.publish(res)

    def fine():
        t = torch.ones(1)
        if world > 1:
            dist.all_reduce(t)
        return {"ms_per_step": 1.0, "tokens_per_s": 1000.0 * float(t.item())}

    def hangs():
        if os.environ.get("SCONE_SELFTEST_DIE_RANK") == str(rank):
            os._exit(7)                            # a rank that crashes in the middle of the record
        if rank != 0 and args.selftest == "hang":
            time.sleep(3600)                       # never joins: rank 0 waits in the collective for ever
        return fine()

    def mutate():                                  # keeps changing the record while the watchdog may be printing it
        i = 0
        while True:
            with line.lock:
                rec.setdefault("noise", {})[f"k{i % 64}"] = i
            i += 1
            time.sleep(0.0005)
    threading.Thread(target=mutate, daemon=True).start()
    run_stages(rec, line, watchdog, budget, args.stage_limit, [("fine", fine), ("second", hangs), ("third", fine)], 500.0)
    line.emit()
    if world > 1:
        watchdog.arm("final barrier", 20.0)
        dist.barrier()
        dist.destroy_process_group()
    sys.stdout.flush()
    os._exit(0)


# ----------------------------------------------------------------------------------------------------------------
def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args, _T_PROCESS_START))
    if args.selftest:
        selftest_main(args)
    import numpy as np
    import torch
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no HIP device visible)")
    budget = Budget(args.time_budget, _T_PROCESS_START)
    line = Line(rank)
    _LINE[0] = line
    watchdog = Watchdog(budget, line, rank)
    watchdog.start()
    # Rehearsal knobs (never set by the driver): SCONE_DIST_BACKEND=gloo + SCONE_ONE_DEVICE=1 let several
    # ranks share ONE GPU so that the N > 1 code path can be exercised on a 1-GPU box (RCCL refuses two
    # ranks on one device).  Numbers from such a run are not scaling results.
    backend = os.environ.get("SCONE_DIST_BACKEND", "nccl")
    if os.environ.get("SCONE_ONE_DEVICE") == "1":
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1 or args.force_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        if rank == 0 and world > 1:
            os.environ.setdefault("SCONE_DIST_TRACE", "1")     # one stderr line per collective of the sharded record (rank 0)
        if backend == "nccl":
            # RCCL's kernels compete with a lookup grid that fills the chip: ask for a high-priority stream for them
            try:
                opts = dist.ProcessGroupNCCL.Options(is_high_priority_stream=True)
                dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank), pg_options=opts)
                NCCL_HIGH_PRIORITY[0] = True
            except Exception as e:                       # an older / different binding: the default stream priority
                sys.stderr.write(f"bench.py: high-priority RCCL stream not available ({e!r}); default priority\n")
                if not dist.is_initialized():
                    dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)

    from scone_amd import EmbeddingCache, NGramExtractor
    from scone_amd import synthetic as S
    from scone_amd.hip_backend import format_code

    # ---- N > 1, before anything that can hang or cost minutes: can this world do a collective at all?
    world_sanity = None
    if dist is not None and world > 1:
        watchdog.arm("world_sanity", min(90.0, max(budget.remaining() - 30.0, 10.0)))
        cdev0 = "cuda" if backend == "nccl" else "cpu"
        t_s = time.perf_counter()
        mine = torch.full((256,), float(rank + 1), dtype=torch.float32, device=cdev0)
        got = torch.empty(256 * world, dtype=torch.float32, device=cdev0)
        dist.all_gather_into_tensor(got, mine)
        sane = bool(torch.equal(got.view(world, 256)[:, 0].cpu(), torch.arange(1, world + 1, dtype=torch.float32)))
        watchdog.disarm()
        world_sanity = {"all_gather_1KB_per_rank_ok": sane, "seconds": time.perf_counter() - t_s, "world_size": dist.get_world_size(),
                        "device_count": torch.cuda.device_count(), "backend": backend,
                        "rccl_version": _rccl_version() if backend == "nccl" else None}
        if rank == 0:
            sys.stderr.write(f"bench.py: world_sanity {'ok' if sane else 'FAILED'}: {world_sanity['world_size']} ranks, "
                             f"{world_sanity['device_count']} devices visible, backend {backend}, RCCL {world_sanity['rccl_version']}, "
                             f"first collective {world_sanity['seconds']:.1f} s\n")
            sys.stderr.flush()
        if not sane:
            raise SystemExit("bench.py: world sanity failed: a 1 KB all-gather returned the wrong ranks' data")

    d, N, B, T = args.dim, args.rows, args.batch, args.seq
    vocab, max_n, seed, base_scale = args.vocab, 3, 7, 0.02 / 127
    if not 3 <= vocab <= 262144:
        raise SystemExit("--vocab must be in [3, 262144]")
    if args.keygen == "structured":
        try:
            S.check_structured_vocab(vocab)          # a vocabulary that shares a factor with a multiplier has duplicate keys
        except ValueError as e:
            raise SystemExit(f"--vocab: {e}")
    vocab_obj, keys, lens = make_vocabulary(N, args.keygen, max_n, vocab=vocab)
    ex = vocab_obj
    kw_rows = {"n_rows": N} if keys is None else {}

    sharded = args.table_mode == "sharded" and dist is not None
    emu = None
    if args.shard_of:
        r_, w_ = (int(x) for x in args.shard_of.split("/"))
        emu = (r_, w_)
    if emu is not None:
        from scone_amd.distributed import ShardedEmbeddingCache
        cache = ShardedEmbeddingCache.from_synthetic(ex, d, table_format=args.format, seed=seed, base_scale=base_scale,
                                                     rank=emu[0], world=emu[1], **kw_rows)
        stream_seed = 1234
    elif sharded:
        from scone_amd.distributed import ShardedEmbeddingCache
        cache = ShardedEmbeddingCache.from_synthetic(ex, d, table_format=args.format, seed=seed,
                                                     base_scale=base_scale, rank=rank, world=world,
                                                     replicated_rows=args.replicated_rows, **kw_rows)
        stream_seed = 1234            # every rank embeds the same batch; rows are sharded
    else:
        cache = EmbeddingCache.from_synthetic(ex, d, table_format=args.format, seed=seed, base_scale=base_scale,
                                              placement=args.placement, hot_rows=args.hot_rows,
                                              stage_tokens=args.stage_tokens, cache_rows=args.cache_rows, **kw_rows)
        stream_seed = 1234 + rank     # every rank embeds its own batches
    # A DIFFERENT batch every step (round 5; rounds 1-4 re-used one batch -- 0.35 GB of rows + 72 MB of wte rows against a
    # 256-MB Infinity Cache: part of step i's working set was still resident for step i + 1, in the timed run and in the
    # counter passes alike; a pinned-host table behind the prefetch pipeline would even be served from its HBM cache).
    # All batches are generated before the timed region; the byte counts below are those of the first one (the batches are
    # statistically identical).  The sharded / shard-emulation modes keep their one batch (frozen this round).
    rotated = emu is None and not sharded and not args.same_batch
    n_batches = min(args.steps + args.warmup, MAX_DISTINCT_BATCHES) if rotated else 1
    tok_np, batches = make_batches(vocab_obj, keys, lens, args.stream, B, T, stream_seed, max(n_batches, 1))
    tok = batches[0]
    staged = args.placement == "pinned_host" and args.stage_tokens > 0
    prefetch = emu is None and not sharded and (args.prefetch == "on" or (args.prefetch == "auto" and staged))
    g = torch.Generator(device="cuda").manual_seed(5)
    wte = (torch.randn(vocab, d, generator=g, device="cuda") * 0.02).half()
    wpe = (torch.randn(1024, d, generator=g, device="cuda") * 0.01).half()
    # The output buffer is re-used by every step; the kernel's time follows its physical placement (0.616 ... 0.657 ms over five
    # allocations of one process, whatever the table: profiles/r06m), so it is chosen by measurement before the timed region:
    # `--out-candidates` allocations, 5 timed lookups each, the fastest kept (EmbeddingCache.alloc_output; the line says so and
    # prints every candidate's time).  1 = a plain allocation.
    out_report = None
    if args.out_candidates > 1 and emu is None and not sharded and args.placement == "hbm" and hasattr(cache, "alloc_output"):
        out, out_report = cache.alloc_output(batches[0], wte=wte, wpe=wpe, candidates=args.out_candidates)
    else:
        out = torch.empty(B, T, d, dtype=torch.float16, device="cuda")

    # workload statistics (outside the timed region)
    table = cache.table
    ntok = B * T
    fmt = format_code(args.format)
    bytes_per_launch, bytes_compulsory, sum_k, k_hist, n_rows_distinct, n_tok_distinct = workload_bytes(table, tok, fmt, d)

    def step():
        if emu is not None:
            # this shard's local work only: partial sums over owned rows, then finalise 1/W of the tokens
            partial, counts = table.embed_partial(tok)
            per = (ntok + emu[1] - 1) // emu[1]
            a0 = min(emu[0] * per, ntok)
            b0 = min(a0 + per, ntok)
            table.finalize(partial[a0:b0], counts[a0:b0], tok, a0, b0, wte=wte, wpe=wpe, out_dtype=torch.float16,
                           out=out.view(-1, d)[a0:b0])
        else:
            cache.embed_tokens(tok, wte=wte, wpe=wpe, exchange=args.exchange, gather_output=not args.no_gather_output)

    def sync():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    if emu is not None or sharded:
        dt, n_launch, kern_ms, samples = measure_lookup(table, step, tok, ntok, args.steps, args.warmup, sync)
    else:
        dt, n_launch, kern_ms, samples = lookup_loop(cache, batches, wte, wpe, out, args.steps, args.warmup, sync, prefetch)
        cache.embed_tokens(tok, wte=wte, wpe=wpe, out=out)       # `out` is compared with the oracle on the FIRST batch further down
        torch.cuda.synchronize()
    if dist is not None:
        tmax = torch.tensor([dt], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
    line.headline_done = True         # from here on a time-out still leaves a valid line (exit status 0)

    units = ntok * args.steps * (1 if sharded else world)
    value = units / dt
    res = None
    if rank == 0:
        # launches per step: 1, or one per chunk for the staged pinned-host lookup.  The roofline prices the kernel
        # time of a whole STEP against the step's bytes.  Without a timed launch (sharded path: other kernels) fall
        # back to the whole step so that the line stays well-formed
        per_step = max(1, n_launch // max(args.steps, 1)) if n_launch else 1
        step_kernel_ms = kern_ms / args.steps if n_launch else dt / args.steps * 1e3
        in_hbm = args.placement == "hbm"
        sig = workload_sig(args.format, d, N, B, T, args.stream, args.placement, args.keygen, rotated=n_batches > 1, vocab=vocab,
                           extra=("-sharded" if sharded else "") + (f"-shard{args.shard_of}" if emu else "")
                           + (f"-hot{args.hot_rows}-stage{args.stage_tokens}" if args.placement != "hbm" else ""))
        res = {
            "metric": "f-gram embed tokens/sec (1M-row INT8 table @ d=768)" if (N, d, args.format) == (1_000_000, 768, "int8")
                      else f"f-gram embed tokens/sec ({N}-row {args.format} table @ d={d})",
            "value": value, "unit": "tokens/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True,
            "scaling": "strong" if sharded else "weak", "vs_baseline": None,
            "dtype": "f32",       # the arithmetic type of the path: rows dequantised, summed and combined in fp32
            "table_format": args.format, "out_dtype": "f16",
            "data": "synthetic", "workload_sig": sig,
            "config": {
                "workload": f"{N}-row {args.format} f-gram table d={d} max_n={max_n} vocab={vocab} in "
                            f"{'HBM' if args.placement == 'hbm' else 'pinned host DRAM'}; S_{args.stream} stream, "
                            f"{B}x{T} tokens/step/rank; fused match+gather+dequant+mean+wte+wpe, fp16 out",
                "tokens_per_step_per_rank": ntok, "mean_hits_per_token": sum_k / ntok, "hits_histogram_K0_6": k_hist[:7],
                "different_batch_every_step": n_batches > 1, "distinct_batches": n_batches,
                "next_batch_announced": bool(prefetch),
                "output_buffer": ({"how": "fastest of N candidate allocations, 5 timed lookups each, before the timed region "
                                          "(EmbeddingCache.alloc_output): the kernel's time follows the buffer's physical placement",
                                   **out_report} if out_report else {"how": "one plain allocation", "candidates": 1}),
                "loop": ("scone_embed(batch i) then scone_embed_prefetch(batch i + 1): the next batch's first chunks are prepared on the "
                         "handle's side streams beside this batch's last lookups" if prefetch else
                         "scone_embed(batch i): k_match_ell, then the gather kernel, on one stream (running the next batch's match on a "
                         "side stream beside this gather was built and measured 1-19 % slower at every batch size: profiles/r05b, r05c)"),
                "distinct_table_rows_per_launch": n_rows_distinct, "distinct_wte_rows_per_launch": n_tok_distinct,
                "parallelism": (f"shard {args.shard_of} of a row-sharded table, local work only (no exchange)" if emu else
                                (("row-sharded table, RCCL all-to-all of quantised rows"
                                  if (args.exchange == "rows" or (args.exchange == "auto" and args.no_gather_output)) else
                                  "row-sharded table, RCCL all-gather of the distinct quantised rows, whole batch reduced on every rank"
                                  if args.exchange in ("gather_rows", "auto") else
                                  "row-sharded table, RCCL reduce-scatter of fp32 partial sums")
                                 + (", every rank keeps its slice" if args.no_gather_output else
                                    ("" if args.exchange in ("gather_rows", "auto") else " + all-gather of the output"))
                                 + f", replicated head {args.replicated_rows} rows") if sharded
                                else f"replicated table, tokens sharded over {world} rank(s), no collective"),
            },
            "roofline": roofline_block(sig, bytes_per_launch, bytes_compulsory, step_kernel_ms, samples, per_step, n_launch, in_hbm,
                                       kernel=None if n_launch else "whole step (sharded path: match + pack + RCCL + gather)"),
            "time_budget_s": budget.seconds,
        }
        res["roofline"]["step_minus_kernel_us"] = (dt / args.steps * 1e3 - step_kernel_ms) * 1e3
        if world_sanity is not None:
            res["world_sanity"] = world_sanity
        if n_launch and not sharded and emu is None and args.placement == "hbm":
            # the match's share of the step: ms_per_step - avg_kernel_ms (k_match_ell + the gap between the two launches); and a
            # contrast on the same table, rank 0 alone (local synchronisation only): ONE batch repeated -- what the Infinity
            # Cache carries from step to step (rounds 1-4 measured the headline and its counter passes this way)
            try:
                local_sync = torch.cuda.synchronize
                cs = max(10, min(args.steps, 20))

                def contrast(bs, pf):
                    dt_c, nl_c, km_c, sm_c = lookup_loop(cache, bs, wte, wpe, out, cs, 2, local_sync, pf)
                    return {"ms_per_step": dt_c / cs * 1e3, "avg_kernel_ms": km_c / max(nl_c, 1), "kernel_ms": kernel_stats(sm_c, 1),
                            "tokens_per_s": ntok * cs / dt_c, "step_minus_kernel_us": (dt_c / cs * 1e3 - km_c / max(nl_c, 1)) * 1e3,
                            "steps": cs, "distinct_batches": len(bs)}
                res["roofline"]["match_us"] = res["roofline"]["step_minus_kernel_us"]
                res["roofline"]["match_us_kind"] = "ms_per_step - avg_kernel_ms: k_match_ell + the gap between the two launches"
                if n_batches > 1 and not args.quick:     # (--quick runs sit under the counter passes: every launch a fresh batch)
                    res["roofline"]["same_batch"] = contrast(batches[:1], prefetch)
                cache.embed_tokens(tok, wte=wte, wpe=wpe, out=out)       # (`out` = the first batch again, for the oracle check)
                torch.cuda.synchronize()
            except Exception as e:
                res["roofline"]["match_us"] = None
                res["roofline"]["contrast_error"] = repr(e)
        line.publish(res)
    gpu_out_for_check = out
    # ---- the reference's own sizes on the headline table: the latency regime (rank 0 of an N = 1 job, < 5 s)
    if rank == 0 and world == 1 and not sharded and emu is None and not args.no_latency and args.placement == "hbm" \
            and budget.remaining() > 60.0:
        watchdog.arm("latency", 45.0)
        try:
            lat = latency_block(cache, wte, wpe, d)
        except Exception as e:
            lat = {"error": repr(e)}
        watchdog.disarm()
        line.set(res, "latency", lat)
        try:
            cache.embed_tokens(tok, wte=wte, wpe=wpe, out=out)       # (`out` = the first batch again, for the oracle check)
            torch.cuda.synchronize()
        except Exception as e:
            line.set(res, "latency_restore_error", repr(e))
    # ---- the cache-defeating variant, the CPU baselines, the sharded record: outside the timed region -----------
    if rank == 0 and world == 1 and not sharded and emu is None and not args.no_hbm_variant and args.placement == "hbm" \
            and budget.remaining() > 150.0:
        watchdog.arm("roofline.hbm_variant", min(150.0, budget.remaining() - 60.0))
        try:
            hv = hbm_variant(args, wte, wpe, sync, prefetch, check=cpu_baseline_spot_check, out_candidates=args.out_candidates)
            with line.lock:
                res["roofline"]["hbm_variant"] = hv
                # the cache-defeating variant's bracket, lifted to the top of the block: what really touches HBM is at least
                # `_lower` (compulsory bytes) and at most `_upper` (bytes that left L2, Infinity-Cache hits included) of the peak
                res["roofline"]["hbm_variant_frac_lower"] = hv["hbm_frac"]
                res["roofline"]["hbm_variant_frac_upper"] = hv["traffic_frac"]
            # ... and the same table over a 262,144-token vocabulary (round 5): 0.6 GB of token-indexed rows (wte + unigram rows)
            # per launch, 2.3x the Infinity Cache, each referenced ~4 times instead of ~21 -- most of what the 50,257-token variant
            # re-reads out of the Infinity Cache must now come from HBM.  If the rate through the L2-miss path stays where it
            # was, that path (not HBM, not the Infinity Cache) is what bounds the kernel; the compulsory fraction of THIS
            # variant is the tightest lower bound on HBM utilisation the line holds
            if budget.remaining() > 120.0:
                watchdog.arm("roofline.mall_variant", min(120.0, budget.remaining() - 60.0))     # a stage of its own (two 10M-row builds)
                mv = config_record("mall_variant", args.format, args.dim, 10_000_000, "structured", "uniform", args.batch, args.seq,
                                   max(10, min(args.steps, 30)), 3, sync, prefetch, check=cpu_baseline_spot_check, vocab=262144,
                                   out_candidates=args.out_candidates)
                with line.lock:
                    res["roofline"]["mall_variant"] = mv
        except Exception as e:
            line.set(res["roofline"], "hbm_variant" if "hbm_variant" not in res["roofline"] else "mall_variant", {"error": repr(e)})
        watchdog.disarm()
    if rank == 0 and not args.no_cpu_baseline and not sharded and emu is None:
        short = world > 1                     # N > 1: a 3-second 1-core sample; the all-cores figures are on the N = 1 line
        if budget.remaining() > (40.0 if short else 100.0):
            watchdog.arm("cpu_baseline", 60.0 if short else 120.0)
            try:
                cb = cpu_baseline(args, keys, lens, tok_np, seed, base_scale, gpu_out_for_check, wte, wpe,
                                  seconds=3.0 if short else None, all_cores=not short)
            except Exception as e:      # the baseline is reported, never the product
                cb = {"value": None, "unit": "tokens/s", "cores": 1, "kind": "port", "sample": f"failed: {e!r}"}
            watchdog.disarm()
        else:
            cb = {"value": None, "unit": "tokens/s", "cores": 1, "kind": "port", "sample": "skipped: time budget"}
        line.set(res, "cpu_baseline", cb)
    # ---- the other single-GPU configs of BASELINE.json, each measured like the headline (rank 0 of an N = 1 job)
    if rank == 0 and world == 1 and not sharded and emu is None and not args.no_configs and args.placement == "hbm":
        cfgs = {}
        line.set(res, "configs", cfgs)
        if budget.remaining() > 140.0:                 # C1: the reference's own CPU-runnable case, through the GPU path (~15 s)
            watchdog.arm("configs.C1_fp32_100K_d768_from_fit", 60.0)
            try:
                c = c1_record(c1_oracle_leg)
            except Exception as e:
                c = {"error": repr(e)}
                torch.cuda.empty_cache()
            watchdog.disarm()
            line.set(cfgs, "C1_fp32_100K_d768_from_fit", c)
        else:
            line.set(cfgs, "C1_fp32_100K_d768_from_fit", {"skipped": f"time budget: {budget.remaining():.0f} s left"})
        plan = [("C2_fp16_1M_d768", "fp16", 768, 1_000_000, "zipf", 20.0),
                ("C3_int8_10M_d1024", "int8", 1024, 10_000_000, "zipf_gpu", 45.0),
                ("C4_int4_100M_d1024_in_hbm", "int4", 1024, 100_000_000, "structured", 75.0)]
        for cname, cfmt, cd, cN, ckg, need_s in plan:
            # (time: what the stages after this one need -- the pinned-host record ~60 s -- stays reserved)
            if budget.remaining() < need_s + 100.0:
                line.set(cfgs, cname, {"skipped": f"time budget: {budget.remaining():.0f} s left"})
                continue
            free_b, _ = torch.cuda.mem_get_info()
            if free_b < cN * 620 + 20e9:
                line.set(cfgs, cname, {"skipped": f"needs {(cN * 620 + 20e9) / 1e9:.0f} GB of HBM ({free_b / 1e9:.0f} GB free)"})
                continue
            watchdog.arm(f"configs.{cname}", min(need_s + 60.0, budget.remaining() - 30.0))
            try:
                reuse = (vocab_obj, keys, lens) if (cN, ckg) == (N, args.keygen) else None
                c = config_record(cname, cfmt, cd, cN, ckg, "uniform", B, T, args.configs_steps, 3, lambda: torch.cuda.synchronize(),
                                  prefetch, vocab_cache=reuse, wte=wte if cd == d else None, wpe=wpe if cd == d else None,
                                  check=cpu_baseline_spot_check, out_candidates=args.out_candidates)
            except Exception as e:
                c = {"error": repr(e)}
                torch.cuda.empty_cache()
            watchdog.disarm()
            line.set(cfgs, cname, c)
    if not args.no_sharded_record and not sharded and emu is None:
        del cache, table, out, gpu_out_for_check
        torch.cuda.empty_cache()
        rec = {}
        if rank == 0:
            line.set(res, "sharded", rec)
        try:
            if world > 1:
                sharded_record(args, dist, rank, world, backend, sync, rec, line, watchdog, budget)
            elif budget.remaining() > 90.0:
                watchdog.arm("sharded.n1_pinned_host", min(240.0, budget.remaining() - 20.0))
                n1, n1z = pinned_baseline(args, sync, zipf_too=True)
                watchdog.disarm()
                with line.lock:
                    rec["n1_pinned_host"] = n1
                    rec["n1_pinned_host_zipf"] = n1z
                    rec["note"] = ("one GPU: nothing to exchange.  This is the single-GPU alternative for a table that does not fit "
                                   "HBM; the row-sharded record (with this baseline measured again by its rank 0) is printed by the "
                                   "N > 1 lines")
            else:
                line.set(rec, "skipped", f"time budget: {budget.remaining():.0f} s left")
        except Exception as e:
            watchdog.disarm()
            line.set(rec, "error", repr(e))
    line.emit()
    if dist is not None:
        watchdog.arm("final barrier", 30.0)       # the line is out: a rank that never arrives costs 30 s, not the job
        dist.barrier()
        dist.destroy_process_group()
    sys.stdout.flush()
    os._exit(0) if dist is not None else None   # skip RCCL's exit-time stdout chatter after the JSON line


_LINE = [None]      # main()'s Line: an exception after the headline was measured must not cost the line


if __name__ == "__main__":
    try:
        main()
    except Exception as e:
        ln = _LINE[0]
        if ln is None or not ln.headline_done or ln.res is None:
            raise
        import traceback
        traceback.print_exc()
        sys.stderr.write("bench.py: the headline was measured before this error; printing the line as it stands\n")
        ln.emit(incomplete=f"an optional block raised {e!r}; what was measured before it is kept")
        sys.stdout.flush()
        os._exit(0)
