#!/usr/bin/env python3
"""bench.py -- f-gram embed throughput on MI355X (BASELINE.json's metric).

    python bench.py [--gpus N] [--steps K] [--warmup W]

One "step" = one pass of the hot path (n-gram match -> INT8 row gather -> dequantise ->
mean -> + wte + wpe -> fp16 store; `scone_embed`) over one batch of B x T synthetic tokens
already resident in HBM.  Headline workload (N = 1 and every rank at N > 1): 1M-row INT8
f-gram table, d = 768, max_n = 3, GPT-2 vocabulary, S_uniform stream.

N > 1: the 1M-row table fits one GPU, sequences are independent, so the path shards over
tokens -- every rank holds the table and embeds its own batch; no data-path collective;
"scaling": "weak".  (`--table-mode sharded` runs the row-sharded + RCCL exchange path used
for tables larger than one GPU.)

Prints ONE JSON line on rank 0 (contract in the task statement) including `roofline`
(the gather/reduce kernel, HIP-event timed on its launch stream) and `cpu_baseline`
(the line-for-line Python port of the reference loop, oracle/ref_port.py, 1 core).
"""

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
# the host driver only supports dmabuf IPC: must be in the environment BEFORE the HIP runtime starts (RCCL at N > 1)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import numpy as np
import torch

HBM_PEAK_GBPS = 8000.0      # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def parse():
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
    ap.add_argument("--force-dist", action="store_true", help="init torch.distributed even with one rank (tests the N>1 code path)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    return ap.parse_args()


def cpu_baseline(args, keys, lens, tok, seed, base_scale, gpu_out, wte, wpe):
    """Time the reference loop (set-of-tuples match -> dict id map -> torch.stack of fp32 rows
    -> mean -> zero-filled [1,T,d]; n_gram_extractor.py:106-126, embedding_cache.py:113-181,
    engine.py:234-266) on a bounded sample of the same stream, 1 core, and use its output to
    sanity-check the GPU result for the first sequence."""
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
            raise RuntimeError("cpu baseline: int4 host generator not wired")
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
    # courtesy upper bound: the plain-C oracle (oracle/oracle.c), OpenMP over independent sequences on all host
    # cores, on the same sample (vocabulary and table restricted to the f-grams the sample references)
    c_line = None
    try:
        from oracle.c_oracle import COracle
        refs = np.asarray(sorted(cache.embeddings.keys()), dtype=np.int64)
        sub = np.stack([cache.embeddings[int(i)] for i in refs])
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
        "c_oracle_all_cores": c_line,
    }


def read_traffic(sig):
    """HBM bytes per launch from committed rocprofv3 PMC passes (profiles/*.json), if the
    workload signature matches; else None."""
    p = os.path.join(ROOT, "profiles", "hbm_traffic.json")
    try:
        entries = json.load(open(p))
        for e in entries:
            if e.get("workload_sig") == sig:
                return e
    except Exception:
        pass
    return None


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
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
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)

    from scone_amd import EmbeddingCache, NGramExtractor
    from scone_amd import synthetic as S
    from scone_amd.hip_backend import format_code, row_bytes

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
    off, ids = table.match_csr(tok)
    counts = (off[1:] - off[:-1]).to(torch.int64)
    sum_k = int(counts.sum().item())
    k_hist = torch.bincount(counts, minlength=7).tolist()
    ntok = B * T
    fmt = format_code(args.format)
    bytes_per_launch = sum_k * row_bytes(fmt, d) + ntok * (d * 2 + d * 2 + 4)
    del off, ids, counts

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

    if hasattr(table, "reserve"):
        table.reserve(ntok)              # workspaces are allocated here, never inside the timed region (even with --warmup 0)
    for _ in range(args.warmup):
        step()
    table.profile_enable(True)
    table.profile_read(reset=True)
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    sync()
    dt = time.perf_counter() - t0
    n_launch, kern_ms = table.profile_read(reset=True)
    table.profile_enable(False)
    if dist is not None:
        tmax = torch.tensor([dt], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())

    units = ntok * args.steps * (1 if sharded else world)
    value = units / dt
    res = None
    if rank == 0:
        # the sharded path launches other kernels (partial + finalise); without a timed launch fall
        # back to the whole step so that the line stays well-formed
        avg_ms = kern_ms / n_launch if n_launch else dt / args.steps * 1e3
        achieved = bytes_per_launch / (avg_ms * 1e-3) / 1e9
        sig = (f"{args.format}-d{d}-N{N}-B{B}-T{T}-{args.stream}-{args.placement}" + ("-sharded" if sharded else "") + (f"-shard{args.shard_of}" if emu else "")
               + (f"-hot{args.hot_rows}-stage{args.stage_tokens}" if args.placement != "hbm" else ""))
        tr = read_traffic(sig)
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
                "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBPS,
                "algorithmic_bytes_per_launch": bytes_per_launch, "avg_kernel_ms": avg_ms, "timed_launches": n_launch,
                "traffic": None if tr is None else tr.get("hbm_bytes_per_launch"),
                "traffic_source": None if tr is None else tr.get("source"),
                # the same launch priced by the bytes that actually left L2 (PMC), not by the algorithmic bytes:
                # adjacent tokens share f-gram rows and hot wte rows hit L2, so this is the lower figure
                "traffic_GBps": None if tr is None else tr["hbm_bytes_per_launch"] / (avg_ms * 1e-3) / 1e9,
                "traffic_frac": None if tr is None else tr["hbm_bytes_per_launch"] / (avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS,
            },
        }
        if not args.no_cpu_baseline and world == 1 and not sharded:
            try:
                res["cpu_baseline"] = cpu_baseline(args, keys, lens, tok_np, seed, base_scale, out, wte, wpe)
            except Exception as e:      # the baseline is reported, never the product
                res["cpu_baseline"] = {"value": None, "unit": "tokens/s", "cores": 1, "kind": "port",
                                       "sample": f"failed: {e!r}"}
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
