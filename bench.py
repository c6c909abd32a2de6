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

Prints ONE JSON line on rank 0 (contract in the task statement) with
  roofline      the gather/reduce kernel, HIP-event timed on its launch stream: algorithmic GB/s and its
                fraction of the 8 TB/s HBM peak, min / median / max launch time, the compulsory-byte lower bound and the
                PMC upper bound on what really came from HBM (`hbm_frac`, `traffic_frac`), and the same figures for a
                cache-defeating variant of the workload (`hbm_variant`: 10M rows, token ids uniform over the vocabulary)
  cpu_baseline  the line-for-line Python port of the reference loop (oracle/ref_port.py), 1 core; beside it the same
                port on all host cores (multiprocessing over sequences) and the plain-C oracle with OpenMP
  sharded       N > 1: the row-sharded path on the C5-shaped workload (INT4 d = 1024, 125M rows per rank, replicated
                index, 1M-token batch) for both exchanges, with the phase split, wire bytes and RCCL facts;
                N = 1: the single-GPU way to serve a table that does not fit HBM (pinned host DRAM, C4-shaped) -- the
                baseline the north-star's ">= 4x at 8 GPUs" is computed against
"""

import argparse
import hashlib
import json
import os
import socket
import subprocess
import sys
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
# the host driver only supports dmabuf IPC: must be in the environment BEFORE the HIP runtime starts (RCCL at N > 1)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

HBM_PEAK_GBPS = 8000.0      # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


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
    ap.add_argument("--keygen", default="zipf", choices=["zipf", "structured"],
                    help="vocabulary generator: seeded Zipf n-grams with de-duplication (default) or the "
                         "distinct-by-construction generator for >= 1e8 rows")
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
    ap.add_argument("--pinned-rows", type=int, default=100_000_000, help="N = 1 sharded baseline: rows of the pinned-host table (C4)")
    ap.add_argument("--force-dist", action="store_true", help="init torch.distributed even with one rank (tests the N>1 code path)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    a = ap.parse_args(argv)
    if a.quick:
        a.no_cpu_baseline = a.no_hbm_variant = a.no_sharded_record = True
    return a


# ----------------------------------------------------------------------------------------------------------------
# `python bench.py --gpus N` without a launcher: start the N ranks here
def self_launch(args) -> int:
    """Spawn N fresh child processes (one rank per GPU) BEFORE this process touches a GPU -- nothing here calls into HIP,
    and no process that has is ever replaced by another program.  Rank 0's stdout is captured; its last JSON line is
    checked (n_gpus == N) and forwarded as this process's single output line."""
    import torch                                   # device_count() does not initialise the GPU on this image
    n = args.gpus
    one_device = os.environ.get("SCONE_ONE_DEVICE") == "1"
    have = torch.cuda.device_count()
    if have < n and not one_device:
        print(f"bench.py: --gpus {n} but only {have} HIP device(s) visible", file=sys.stderr)
        return 2
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, text=True))
    out, _ = procs[0].communicate()
    rcs = [procs[0].returncode]
    for p in procs[1:]:
        try:
            rcs.append(p.wait(timeout=120))
        except subprocess.TimeoutExpired:
            p.kill()
            rcs.append(-9)
    line = None
    for ln in (out or "").splitlines():
        if ln.startswith('{"metric"'):
            line = ln
    if any(rcs) or line is None:
        sys.stderr.write(f"bench.py: ranks exited with {rcs}; rank 0 printed {'no' if line is None else 'a'} result line\n")
        if out:
            sys.stderr.write(out[-2000:])
        return 1
    res = json.loads(line)
    if res.get("n_gpus") != n:
        sys.stderr.write(f"bench.py: --gpus {n} but the result line says n_gpus = {res.get('n_gpus')}\n")
        return 1
    res["launcher"] = f"bench.py started {n} ranks itself (WORLD_SIZE was unset)"
    print(json.dumps(res), flush=True)
    return 0


# ----------------------------------------------------------------------------------------------------------------
def kernel_source_sha() -> str:
    """Hash of the kernel sources: a committed PMC traffic figure is only quoted for the code it was measured on."""
    h = hashlib.sha256()
    d = os.path.join(ROOT, "scone_amd", "csrc")
    for f in sorted(os.listdir(d)):
        if f.endswith((".hip", ".h")):
            h.update(f.encode())
            h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]


def read_traffic(sig):
    """Bytes that left L2 per launch from committed rocprofv3 PMC passes (profiles/hbm_traffic.json) if the workload
    signature matches; (entry, stale) -- stale when the kernels have changed since the passes were taken."""
    p = os.path.join(ROOT, "profiles", "hbm_traffic.json")
    try:
        for e in json.load(open(p)):
            if e.get("workload_sig") == sig:
                return e, e.get("kernel_source_sha") != kernel_source_sha()
    except Exception:
        pass
    return None, False


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


def cpu_baseline(args, keys, lens, tok, seed, base_scale, gpu_out, wte, wpe):
    """Time the reference loop (set-of-tuples match -> dict id map -> torch.stack of fp32 rows
    -> mean -> zero-filled [1,T,d]; n_gram_extractor.py:106-126, embedding_cache.py:113-181,
    engine.py:234-266) on a bounded sample of the same stream, 1 core, and use its output to
    sanity-check the GPU result for the first sequence."""
    import numpy as np
    import torch
    if keys.shape[0] > 20_000_000:
        raise RuntimeError("cpu baseline skipped: a Python dict of > 2e7 f-grams does not fit the time budget")
    from oracle import ref_port as R
    torch.set_num_threads(1)
    d = args.dim
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

    seqs = [tok[b].tolist() for b in range(min(tok.shape[0], 128))]
    load_rows(seqs)                       # host copies of the rows the sample touches (not timed)
    first = R.aggregate(cache, seqs[0], d)
    # whole passes over the sample until ~cpu_seconds of CPU work have been timed
    done, dt = 0, 0.0
    t0 = time.perf_counter()
    while dt < args.cpu_seconds:
        for s in seqs:
            R.aggregate(cache, s, d)
        done += len(seqs)
        dt = time.perf_counter() - t0
    nseq = done
    refs = np.asarray(sorted(cache.embeddings.keys()), dtype=np.int64)
    sub = np.stack([cache.embeddings[int(i)] for i in refs])
    # courtesy upper bound 1: the SAME Python port on all host cores, multiprocessing over
    # independent sequences (spawned workers -- this process holds a GPU context -- each with the sample's vocabulary)
    py_all = None
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
    c_line = None
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
    # parity spot check of the GPU output (first sequence) against the oracle
    ref = R.combine(torch.from_numpy(tok[:1]), first, wte.float().cpu(), wpe.float().cpu()).numpy()
    err = float(np.abs(gpu_out[:1].float().cpu().numpy() - ref).max() / np.abs(ref).max())
    return {
        "value": nseq * tok.shape[1] / dt, "unit": "tokens/s", "cores": 1, "kind": "port",
        "sample": f"{nseq} sequences x {tok.shape[1]} tokens ({len(seqs)} distinct sequences of the same stream, repeated) "
                  f"({dt:.1f} s; oracle/ref_port.py aggregate(), python {sys.version_info.major}.{sys.version_info.minor}, "
                  f"torch {torch.__version__}, host cpus {os.cpu_count()})",
        "gpu_vs_oracle_max_rel_err_seq0": err,
        "python_all_cores": py_all,
        "c_oracle_all_cores": c_line,
    }


# ----------------------------------------------------------------------------------------------------------------
def measure_lookup(table, embed, tok, ntok, steps, warmup, sync):
    """W untimed + K timed passes of `embed`; returns (seconds, launches, kernel-ms samples)."""
    if hasattr(table, "reserve"):
        table.reserve(ntok)              # workspaces are allocated here, never inside the timed region (even with --warmup 0)
    for _ in range(warmup):
        embed()
    table.profile_enable(True)
    table.profile_read(reset=True)
    sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        embed()
    sync()
    dt = time.perf_counter() - t0
    samples = table.profile_samples()
    n_launch, kern_ms = table.profile_read(reset=True)
    table.profile_enable(False)
    return dt, n_launch, kern_ms, samples


def workload_bytes(table, tok, fmt, d, out_bytes=2, base_bytes=2):
    """(algorithmic bytes per launch [SURVEY 8d: every reference counted], compulsory bytes per launch [every DISTINCT
    table row and wte row once + the output + ids: a lower bound on what must come from / go to HBM when nothing
    survives in cache between launches], sum K, K histogram)."""
    import torch
    from scone_amd.hip_backend import row_bytes
    off, ids = table.match_csr(tok)
    counts = (off[1:] - off[:-1]).to(torch.int64)
    sum_k = int(counts.sum().item())
    k_hist = torch.bincount(counts, minlength=7).tolist()
    ntok = tok.numel()
    algorithmic = sum_k * row_bytes(fmt, d) + ntok * (d * out_bytes + d * base_bytes + 4)
    n_rows_distinct = int(torch.unique(ids).numel())
    n_tok_distinct = int(torch.unique(tok).numel())
    compulsory = n_rows_distinct * row_bytes(fmt, d) + n_tok_distinct * d * base_bytes + ntok * (d * out_bytes + 4)
    return algorithmic, compulsory, sum_k, k_hist, n_rows_distinct, n_tok_distinct


def kernel_stats(samples, per_step):
    """min / median / max of the per-STEP kernel time (a staged lookup launches the kernel once per chunk: its chunks are
    summed per step)."""
    import numpy as np
    s = np.asarray(samples, dtype=np.float64)
    if s.size == 0:
        return None
    if per_step > 1 and s.size % per_step == 0:
        s = s.reshape(-1, per_step).sum(axis=1)
    return {"min": float(s.min()), "median": float(np.median(s)), "max": float(s.max()), "n": int(s.size)}


def hbm_variant(args, wte, wpe, sync):
    """The headline's format and dim on a workload that defeats the caches: 10M rows (7.7 GB of INT8 d = 768 rows -- 30x
    the Infinity Cache), structured vocabulary (token ids uniform over the 50,257-word vocabulary, one bigram / trigram
    row per window, each referenced by the 2-3 adjacent tokens it covers and by nothing else in the launch)."""
    import torch
    from scone_amd import EmbeddingCache
    from scone_amd import synthetic as S
    from scone_amd.hip_backend import format_code
    d, B, T, N = args.dim, args.batch, args.seq, 10_000_000
    vocab = S.StructuredVocab(N)
    cache = EmbeddingCache.from_synthetic(vocab, d, table_format=args.format, seed=7, base_scale=0.02 / 127, n_rows=N)
    tok = torch.from_numpy(S.stream_uniform_ids(vocab, None, B, T, 4321)).to("cuda", torch.int32)
    out = torch.empty(B, T, d, dtype=torch.float16, device="cuda")
    table = cache.table
    alg, comp, sum_k, k_hist, nr, nt = workload_bytes(table, tok, format_code(args.format), d)
    steps = max(10, min(args.steps, 30))
    dt, n_launch, kern_ms, samples = measure_lookup(table, lambda: cache.embed_tokens(tok, wte=wte, wpe=wpe, out=out), tok,
                                                    B * T, steps, 3, sync)
    avg_ms = kern_ms / n_launch
    sig = f"{args.format}-d{d}-N{N}-B{B}-T{T}-uniform-hbm-structured"
    tr, stale = read_traffic(sig)
    res = {
        "workload": f"{N}-row {args.format} table d={d}, structured vocabulary (token ids uniform over the vocabulary), S_uniform, "
                    f"{B}x{T} tokens; {nr} distinct table rows and {nt} distinct wte rows per launch",
        "workload_sig": sig, "mean_hits_per_token": sum_k / (B * T),
        "avg_kernel_ms": avg_ms, "kernel_ms": kernel_stats(samples, 1), "tokens_per_s": B * T * steps / dt,
        "algorithmic_bytes_per_launch": alg, "algorithmic_GBps": alg / avg_ms / 1e6, "algorithmic_frac": alg / avg_ms / 1e6 / HBM_PEAK_GBPS,
        "hbm_bytes_compulsory": comp, "hbm_GBps": comp / avg_ms / 1e6, "hbm_frac": comp / avg_ms / 1e6 / HBM_PEAK_GBPS,
        "traffic": None if (tr is None or stale) else tr["hbm_bytes_per_launch"],
        "traffic_frac": None if (tr is None or stale) else tr["hbm_bytes_per_launch"] / avg_ms / 1e6 / HBM_PEAK_GBPS,
    }
    del cache, table, tok, out
    torch.cuda.empty_cache()
    return res


def pinned_baseline(args, sync):
    """N = 1: how ONE GPU serves a table that does not fit its HBM -- rows in pinned host DRAM read in place over PCIe
    (BASELINE config C4: 100M rows INT4 d = 1024 = 52.8 GB, first 1M rows hot in HBM), on the batch shape of the
    sharded record.  PCIe-bound, so the rate barely depends on the table's size; this is what ">= 4x at 8 GPUs vs 1 GPU
    on the 1B-row sharded table" is computed against."""
    import psutil
    import torch
    from scone_amd import EmbeddingCache
    from scone_amd import synthetic as S
    N, d, B, T = args.pinned_rows, 1024, 2048, 512
    need = N * 512 + 8e9
    avail = psutil.virtual_memory().available
    for f_lim, f_use in (("/sys/fs/cgroup/memory.max", "/sys/fs/cgroup/memory.current"),
                         ("/sys/fs/cgroup/memory/memory.limit_in_bytes", "/sys/fs/cgroup/memory/memory.usage_in_bytes")):
        try:                                            # a container's own limit counts, not only the host's free memory
            lim = open(f_lim).read().strip()
            if lim != "max":
                avail = min(avail, int(lim) - int(open(f_use).read().strip()))
        except (OSError, ValueError):
            pass
    if avail < need:
        return {"value": None, "skipped": f"needs {need / 1e9:.0f} GB of host memory for the pinned table ({avail / 1e9:.0f} GB available)"}
    vocab = S.StructuredVocab(N)
    cache = EmbeddingCache.from_synthetic(vocab, d, table_format="int4", seed=7, base_scale=0.02 / 127, n_rows=N,
                                          placement="pinned_host", hot_rows=1_000_000)
    tok = torch.from_numpy(S.stream_uniform_ids(vocab, None, B, T, 1234)).to("cuda", torch.int32)
    g = torch.Generator(device="cuda").manual_seed(5)
    wte = (torch.randn(S.GPT2_VOCAB, d, generator=g, device="cuda") * 0.02).half()
    wpe = (torch.randn(1024, d, generator=g, device="cuda") * 0.01).half()
    out = torch.empty(B, T, d, dtype=torch.float16, device="cuda")
    cache.table.reserve(B * T)
    cache.embed_tokens(tok, wte=wte, wpe=wpe, out=out)
    sync()
    steps = 4
    t0 = time.perf_counter()
    for _ in range(steps):
        cache.embed_tokens(tok, wte=wte, wpe=wpe, out=out)
    sync()
    dt = time.perf_counter() - t0
    res = {"value": B * T * steps / dt, "unit": "tokens/s", "ms_per_step": dt / steps * 1e3, "steps": steps,
           "workload": f"{N}-row int4 table d={d} in pinned host DRAM (rows read in place over PCIe), first 1000000 rows in HBM, "
                       f"structured vocabulary, S_uniform, {B}x{T} tokens/step",
           "bound": "PCIe Gen5 x16 (~64 GB/s)"}
    del cache, tok, out, wte, wpe
    torch.cuda.empty_cache()
    return res


def _rccl_version():
    try:
        import torch
        return ".".join(str(x) for x in torch.cuda.nccl.version())
    except Exception as e:
        return f"unknown ({e!r})"


def sharded_record(args, dist, rank, world, backend, sync, rec=None):
    """N > 1: the row-sharded path on the C5-shaped workload -- INT4 d = 1024, `rows_per_rank` x N rows (1e9 at N = 8),
    replicated index built from keys generated on the GPU, every rank its own contiguous row range generated on its GPU,
    replicated head = the unigram rows, ONE 1M-token S_uniform batch that every rank passes in -- for the two exchanges
    that leave the whole [B, T, d] output on every rank:
      rows+all_gather   all-to-all of the quantised rows each slice needs, rank r reduces slice r, all-gather of the
                        finished fp16 vectors (the north-star's wording)
      gather_rows       all-gather of the DISTINCT quantised rows the batch references, every rank reduces the whole batch;
                        pipelined over `gather_chunks` chunks of sequences (gather_rows_one_shot: the same in one piece);
                        the records travel as exact point-to-point ranges (batch_isend_irecv: one RCCL group, each link
                        carries one peer's records) -- gather_rows_padded_all_gather: through all_gather_into_tensor
                        instead, every contribution padded to the largest
    and, for contrast, rows_slices_only: the all-to-all alone, every rank keeps its own slice (a consumer that is
    data-parallel over the same slices needs no more); last, gather_rows_split_phase: the serving-loop form of gather_rows
    (ShardedEmbeddingCache.gather_rows_begin / _finish, one piece, two batches in flight: plan, pack and transfers of step
    s + 1 run on a side stream behind the reduction of step s) -- a throughput figure, a batch's latency is two steps.
    Un-synchronised steps give ms/step; one instrumented step per exchange (device synchronised between phases) gives the
    phase split.  `rec` (optional) is filled in place, so that a caller's watchdog can print what was measured so far."""
    import torch
    from scone_amd import synthetic as S
    from scone_amd.distributed import ShardedEmbeddingCache
    d, B, T = 1024, 2048, 512
    free, total = torch.cuda.mem_get_info()
    fm = torch.tensor([float(free)], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
    dist.all_reduce(fm, op=dist.ReduceOp.MIN)       # every rank must size the table the same way: the tightest GPU decides
    free = float(fm.item())
    per = args.sharded_rows_per_rank
    N = per * world
    cap = 64
    while cap < 2 * N:
        cap <<= 1
    need = per * 544 + cap * 17 + 12e9
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
    rec = {} if rec is None else rec
    rec.update({"workload": f"{N}-row int4 f-gram table d={d} row-sharded over {world} ranks ({per} rows = {per * 528 / 1e9:.1f} GB per rank), "
                       f"replicated {cap}-slot index, replicated head {S.GPT2_VOCAB} rows, structured vocabulary, S_uniform, "
                       f"{B}x{T} tokens/step (the same batch on every rank), whole [B,T,d] fp16 output on every rank",
           "world_size": dist.get_world_size(), "device_count": torch.cuda.device_count(), "backend": backend,
           "rccl_version": _rccl_version() if backend == "nccl" else None,
           "build_s": t_build, "note": note, "exchanges": {}})
    checks = {}
    chunks = cache.gather_chunks
    # order: the plainest collectives first (all_to_all_single, all_gather_into_tensor), the batched point-to-point
    # transport after them -- should one of them hang under RCCL, the watchdog still prints everything measured before it
    for name, kw in (("rows+all_gather", {"exchange": "rows", "gather_output": True}),
                     ("rows_slices_only", {"exchange": "rows", "gather_output": False}),
                     ("gather_rows_padded_all_gather", {"exchange": "gather_rows", "gather_output": True}),
                     ("gather_rows", {"exchange": "gather_rows", "gather_output": True}),
                     ("gather_rows_one_shot", {"exchange": "gather_rows", "gather_output": True})):
        try:
            cache.gather_chunks = 1 if name == "gather_rows_one_shot" else chunks
            # the records travel as exact point-to-point ranges; ..._padded_all_gather: all_gather_into_tensor, every
            # contribution padded to the largest (twice the mean on this workload at 8 ranks)
            cache.gather_transport = "all_gather" if name == "gather_rows_padded_all_gather" else "p2p"
            out = cache.embed_tokens(tok, wte=wte, wpe=wpe, **kw)                  # warm-up (allocations, RCCL channels)
            out = cache.embed_tokens(tok, wte=wte, wpe=wpe, **kw)
            sync()
            t0 = time.perf_counter()
            for _ in range(args.sharded_steps):
                out = cache.embed_tokens(tok, wte=wte, wpe=wpe, **kw)
            sync()
            dt = time.perf_counter() - t0
            tm = torch.tensor([dt], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
            dist.all_reduce(tm, op=dist.ReduceOp.MAX)
            dt = float(tm.item())
            phases = cache.embed_tokens(tok, wte=wte, wpe=wpe, profile=True, **kw)[1]
            ph = torch.tensor([phases[k] for k in sorted(phases)], dtype=torch.float64,
                              device="cuda" if backend == "nccl" else "cpu")
            dist.all_reduce(ph, op=dist.ReduceOp.MAX)                                 # slowest rank per phase
            if kw["gather_output"]:
                checks[name] = float(out.float().abs().sum().item())
            rec["exchanges"][name] = {
                "ms_per_step": dt / args.sharded_steps * 1e3, "tokens_per_s": ntok * args.sharded_steps / dt,
                "steps": args.sharded_steps,
                "phase_ms_slowest_rank": {k: float(v) for k, v in zip(sorted(phases), ph.tolist()) if not k.startswith("bytes")},
                "wire_bytes_received_rank0": int(phases.get("bytes_received", 0)),
                "chunks": cache.gather_chunks if kw["exchange"] == "gather_rows" else None,
                "records_transport": ({"p2p": "batch_isend_irecv, exact ranges", "all_gather": "all_gather_into_tensor, padded"}
                                      [cache.gather_transport] if kw["exchange"] == "gather_rows" else "all_to_all_single"),
            }
        except Exception as e:                                                        # the record never takes the line down
            rec["exchanges"][name] = {"error": repr(e)}
    try:
        cache.gather_chunks = 1
        def loop(n):
            o = None
            tk = cache.gather_rows_begin(tok, tokens_ready=None)         # (the batch has been on the device since the build)
            for i in range(n):
                o = cache.gather_rows_finish(tk, wte=wte, wpe=wpe)      # queues the reduction of step i ...
                tk = cache.gather_rows_begin(tok, tokens_ready=None) if i + 1 < n else None   # ... plan / pack / transfers of step i + 1 overlap it
            return o
        out = loop(3)
        sync()
        t0 = time.perf_counter()
        out = loop(args.sharded_steps)
        sync()
        dt = time.perf_counter() - t0
        tm = torch.tensor([dt], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(tm, op=dist.ReduceOp.MAX)
        dt = float(tm.item())
        rec["exchanges"]["gather_rows_split_phase"] = {
            "ms_per_step": dt / args.sharded_steps * 1e3, "tokens_per_s": ntok * args.sharded_steps / dt,
            "steps": args.sharded_steps, "batches_in_flight": 2, "chunks": 1,
            "records_transport": "batch_isend_irecv, exact ranges",
            "same_output_as_gather_rows": bool(float(out.float().abs().sum().item()) == checks.get("gather_rows")),
        }
    except Exception as e:
        rec["exchanges"]["gather_rows_split_phase"] = {"error": repr(e)}
    cache.gather_chunks = chunks
    cache.gather_transport = "p2p"
    if len(checks) == 4:                                # all bit-identical to the unsharded table, hence to each other
        rec["exchanges_agree"] = bool(len(set(checks.values())) == 1)
    rec["gather_chunks"] = chunks
    rec["n1_baseline"] = ("the N = 1 line's `sharded.n1_pinned_host` (one GPU cannot hold this table: rows in pinned host DRAM, "
                          "PCIe-bound, ~0.24 G tokens/s on MI355X); '>= 4x at 8 GPUs' is tokens_per_s here / that value")
    del cache, tok, wte, wpe
    torch.cuda.empty_cache()
    return rec


# ----------------------------------------------------------------------------------------------------------------
def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args))
    import numpy as np
    import torch
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no HIP device visible)")
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
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)

    from scone_amd import EmbeddingCache, NGramExtractor
    from scone_amd import synthetic as S
    from scone_amd.hip_backend import format_code

    d, N, B, T = args.dim, args.rows, args.batch, args.seq
    vocab, max_n, seed, base_scale = S.GPT2_VOCAB, 3, 7, 0.02 / 127
    keys, lens = (S.make_keys(N, vocab, max_n, seed=11) if args.keygen == "zipf"
                  else S.make_keys_structured(N, vocab, max_n))
    ex = NGramExtractor.from_arrays(keys, lens, max_n=max_n)

    sharded = args.table_mode == "sharded" and dist is not None
    emu = None
    if args.shard_of:
        r_, w_ = (int(x) for x in args.shard_of.split("/"))
        emu = (r_, w_)
    if emu is not None:
        from scone_amd.distributed import ShardedEmbeddingCache
        cache = ShardedEmbeddingCache.from_synthetic(ex, d, table_format=args.format, seed=seed, base_scale=base_scale,
                                                     rank=emu[0], world=emu[1])
        stream_seed = 1234
    elif sharded:
        from scone_amd.distributed import ShardedEmbeddingCache
        cache = ShardedEmbeddingCache.from_synthetic(ex, d, table_format=args.format, seed=seed,
                                                     base_scale=base_scale, rank=rank, world=world,
                                                     replicated_rows=args.replicated_rows)
        stream_seed = 1234            # every rank embeds the same batch; rows are sharded
    else:
        cache = EmbeddingCache.from_synthetic(ex, d, table_format=args.format, seed=seed, base_scale=base_scale,
                                              placement=args.placement, hot_rows=args.hot_rows,
                                              stage_tokens=args.stage_tokens)
        stream_seed = 1234 + rank     # every rank embeds its own batch
    if args.stream == "uniform":
        tok_np = S.stream_uniform_ids(keys, lens, B, T, stream_seed)
    else:
        tok_np = S.stream_zipf(vocab, B, T, stream_seed)
    tok = torch.from_numpy(tok_np).to("cuda", torch.int32)
    g = torch.Generator(device="cuda").manual_seed(5)
    wte = (torch.randn(vocab, d, generator=g, device="cuda") * 0.02).half()
    wpe = (torch.randn(1024, d, generator=g, device="cuda") * 0.01).half()
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
        elif sharded:
            cache.embed_tokens(tok, wte=wte, wpe=wpe, exchange=args.exchange, gather_output=not args.no_gather_output)
        else:
            cache.embed_tokens(tok, wte=wte, wpe=wpe, out=out)

    def sync():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    dt, n_launch, kern_ms, samples = measure_lookup(table, step, tok, ntok, args.steps, args.warmup, sync)
    if dist is not None:
        tmax = torch.tensor([dt], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())

    units = ntok * args.steps * (1 if sharded else world)
    value = units / dt
    res = None
    if rank == 0:
        # launches per step: 1, or one per chunk for the staged pinned-host lookup.  The roofline prices the kernel
        # time of a whole STEP against the step's bytes.  Without a timed launch (sharded path: other kernels) fall
        # back to the whole step so that the line stays well-formed
        per_step = max(1, n_launch // max(args.steps, 1)) if n_launch else 1
        step_kernel_ms = kern_ms / args.steps if n_launch else dt / args.steps * 1e3
        achieved = bytes_per_launch / (step_kernel_ms * 1e-3) / 1e9
        sig = (f"{args.format}-d{d}-N{N}-B{B}-T{T}-{args.stream}-{args.placement}" + ("-sharded" if sharded else "") + (f"-shard{args.shard_of}" if emu else "")
               + (f"-hot{args.hot_rows}-stage{args.stage_tokens}" if args.placement != "hbm" else "")
               + ("-structured" if args.keygen == "structured" else ""))
        tr, stale = read_traffic(sig)
        traffic = None if (tr is None or stale) else tr.get("hbm_bytes_per_launch")
        in_hbm = args.placement == "hbm"
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
            "roofline": {
                "bound": "hbm", "kernel": ("scone_gather::k_embed_wave (gather+dequant+reduce+combine), HIP-event timed" if n_launch
                                           else "whole step (sharded path: match + pack + RCCL + gather)"),
                # SURVEY 8d's figure: every row REFERENCE counted (K_t rows + out + wte + id per token).  Adjacent tokens
                # share f-gram rows and hot wte rows are re-referenced, so part of these bytes is served by L2 / the
                # Infinity Cache: `frac` is the algorithmic rate over the HBM peak, not an HBM utilisation
                "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBPS,
                "frac_kind": "algorithmic bytes (SURVEY.md 8d: every reference) / kernel time / 8 TB/s",
                "algorithmic_above_peak": bool(achieved > HBM_PEAK_GBPS),
                "algorithmic_bytes_per_launch": bytes_per_launch, "avg_kernel_ms": step_kernel_ms,
                "kernel_ms": kernel_stats(samples, per_step), "timed_launches": n_launch, "launches_per_step": per_step,
                # what HBM must at least move: every DISTINCT table row and wte row once + the output + the ids
                # (<= the truth; can never exceed the peak) ...
                "hbm_bytes_compulsory": bytes_compulsory if in_hbm else None,
                "hbm_frac": bytes_compulsory / (step_kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS if in_hbm else None,
                # ... and what left L2 (rocprofv3 PMC passes of THIS code, committed: >= the truth, it includes
                # Infinity-Cache hits); null when the kernels have changed since the passes were taken
                "traffic": traffic,
                "traffic_source": None if tr is None else tr.get("source"),
                "traffic_stale": bool(stale),
                "traffic_GBps": None if traffic is None else traffic / (step_kernel_ms * 1e-3) / 1e9,
                "traffic_frac": None if traffic is None else traffic / (step_kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                "kernel_source_sha": kernel_source_sha(),
            },
        }
    gpu_out_for_check = out
    # ---- the cache-defeating variant, the CPU baselines, the sharded record: outside the timed region -----------
    if rank == 0 and world == 1 and not sharded and emu is None and not args.no_hbm_variant and args.placement == "hbm":
        try:
            hv = hbm_variant(args, wte, wpe, sync)
            res["roofline"]["hbm_variant"] = hv
            # the cache-defeating variant's bracket, lifted to the top of the block: what really touches HBM is at least
            # `_lower` (compulsory bytes) and at most `_upper` (bytes that left L2, Infinity-Cache hits included) of the peak
            res["roofline"]["hbm_variant_frac_lower"] = hv["hbm_frac"]
            res["roofline"]["hbm_variant_frac_upper"] = hv["traffic_frac"]
        except Exception as e:
            res["roofline"]["hbm_variant"] = {"error": repr(e)}
    if rank == 0 and not args.no_cpu_baseline and world == 1 and not sharded:
        try:
            res["cpu_baseline"] = cpu_baseline(args, keys, lens, tok_np, seed, base_scale, gpu_out_for_check, wte, wpe)
        except Exception as e:      # the baseline is reported, never the product
            res["cpu_baseline"] = {"value": None, "unit": "tokens/s", "cores": 1, "kind": "port",
                                   "sample": f"failed: {e!r}"}
    if not args.no_sharded_record and not sharded and emu is None:
        del cache, table, out, gpu_out_for_check
        torch.cuda.empty_cache()
        # a hung collective must not cost the headline: after 10 minutes rank 0 prints what it has and every rank leaves
        # (all with status 0, so that the launcher reports the run as what it is: a measured headline without the record)
        partial = {}
        def bail():
            if rank == 0:
                partial["error"] = "timed out after 600 s; what was measured until then is kept"
                res["sharded"] = partial
                print(json.dumps(res, default=str), flush=True)
            os._exit(0)
        watchdog = threading.Timer(600.0 if rank == 0 else 615.0, bail)
        watchdog.daemon = True
        watchdog.start()
        try:
            if world > 1:
                rec = sharded_record(args, dist, rank, world, backend, sync, partial)
            else:
                rec = {"n1_pinned_host": pinned_baseline(args, sync),
                       "note": "one GPU: nothing to exchange.  This is the single-GPU alternative for a table that does not fit "
                               "HBM; the row-sharded record is printed by the N > 1 lines"}
        except Exception as e:
            rec = {"error": repr(e)}
        watchdog.cancel()
        if rank == 0:
            res["sharded"] = rec
    if rank == 0:
        # RCCL prints its version banner through C stdio; flush it first so that the JSON line is
        # the last thing on stdout
        try:
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
        print(json.dumps(res), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    sys.stdout.flush()
    os._exit(0) if dist is not None else None   # skip RCCL's exit-time stdout chatter after the JSON line


if __name__ == "__main__":
    main()
