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

HBM_PEAK_GBPS = 8000.0      # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
PCIE_PEAK_GBPS = 64.0       # PCIe Gen5 x16, one direction
# xGMI: 7 links per GPU, ~153.6 GB/s each counting both directions (task statement / DESIGN section 6) = 76.8 GB/s into a
# GPU per link; a rank receives over min(W - 1, 7) links at once on the fully connected node
XGMI_LINK_GBPS_PER_DIRECTION = 76.8
T0_ENV = "SCONE_BENCH_T0"   # wall-clock start of the job's FIRST process: children of `self_launch` inherit the deadline


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
    ap.add_argument("--no-configs", action="store_true", help="skip the `configs` block (BASELINE configs C2, C3, C4-in-HBM)")
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
        a.no_cpu_baseline = a.no_hbm_variant = a.no_sharded_record = a.no_configs = True
    return a


# ----------------------------------------------------------------------------------------------------------------
# the time budget, the line, the watchdog
class Budget:
    """One absolute deadline for the whole job (wall clock, shared with the ranks `self_launch` starts)."""

    def __init__(self, seconds: float) -> None:
        self.t0 = float(os.environ.get(T0_ENV) or _T_PROCESS_START)
        self.seconds = float(seconds)
        self.deadline = self.t0 + self.seconds

    def remaining(self) -> float:
        return self.deadline - time.time()

    def used(self) -> float:
        return time.time() - self.t0


LINE_LIMIT = 6000            # characters of the printed line (the driver keeps the TAIL of stdout: round 4's 6.9 KB line survived)


def _sig(x, digits=6):
    """Floats to `digits` significant digits (what the line prints); everything else unchanged."""
    if isinstance(x, float):
        if x != x or x in (float("inf"), float("-inf")):
            return None
        return float(f"{x:.{digits}g}")
    if isinstance(x, dict):
        return {k: _sig(v, digits) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [_sig(v, digits) for v in x]
    return x


def _pick(d, keys):
    return {k: d[k] for k in keys if isinstance(d, dict) and k in d and d[k] is not None}


def _cut(text, n):
    return text if not isinstance(text, str) or len(text) <= n else text[:n - 3] + "..."


def compact_record(res, details_path=None):
    """The ONE printed line: the contract's keys and the figures a reader needs, in at most LINE_LIMIT characters -- the whole
    record (every phase split, workload description, provenance string) goes to `details_path`.  Built with .get everywhere:
    the watchdog may print a record that is only partly filled."""
    rf = res.get("roofline") or {}
    out = _pick(res, ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                      "dtype", "table_format", "out_dtype", "data", "workload_sig"))
    out.setdefault("vs_baseline", None)
    cfg = res.get("config") or {}
    out["config"] = _pick(cfg, ("tokens_per_step_per_rank", "mean_hits_per_token", "different_batch_every_step", "distinct_batches",
                                "distinct_table_rows_per_launch", "distinct_wte_rows_per_launch", "next_batch_announced"))
    out["config"]["workload"] = _cut(cfg.get("workload"), 190)
    out["config"]["parallelism"] = _cut(cfg.get("parallelism"), 110)
    left_l2 = rf.get("traffic") is not None

    def roof(r, extra=()):
        c = _pick(r, ("bound", "limited_by", "achieved", "peak", "unit", "frac", "frac_bytes", "algorithmic_bytes_per_launch",
                      "algorithmic_frac", "avg_kernel_ms", "kernel_ms", "timed_launches", "hbm_bytes_compulsory", "hbm_frac", "traffic",
                      "traffic_stale", "traffic_frac", "kernel_source_sha") + tuple(extra))
        if "frac_kind" in r:
            c["frac_kind"] = ("left L2: 2*FETCH_SIZE+WRITE_SIZE (rocprofv3 PMC, this kernel source)" if r.get("traffic") is not None
                              else "compulsory bytes (no PMC entry)") + " / HIP-event kernel time / 8 TB/s"
        if r.get("traffic_source"):
            c["traffic_source"] = r["traffic_source"].split(":")[0]
        return c
    o_rf = roof(rf, ("match_us", "step_minus_kernel_us"))
    if rf.get("kernel"):
        o_rf["kernel"] = _cut(rf["kernel"], 60)
    if isinstance(rf.get("same_batch"), dict):
        o_rf["same_batch"] = _pick(rf["same_batch"], ("ms_per_step", "avg_kernel_ms", "tokens_per_s", "distinct_batches"))
    hv = rf.get("hbm_variant")
    if isinstance(hv, dict):
        o_rf["hbm_variant"] = _pick(hv, ("tokens_per_s", "avg_kernel_ms", "hbm_frac", "traffic_frac", "algorithmic_frac",
                                         "gpu_vs_oracle_max_rel_err", "error"))
    mv = rf.get("mall_variant")
    if isinstance(mv, dict):
        o_rf["mall_variant"] = {**_pick(mv, ("tokens_per_s", "gpu_vs_oracle_max_rel_err", "error")),
                                **_pick(mv.get("roofline") or {}, ("avg_kernel_ms", "hbm_frac", "traffic_frac", "algorithmic_frac"))}
    out["roofline"] = o_rf
    cb = res.get("cpu_baseline")
    if isinstance(cb, dict):
        o_cb = _pick(cb, ("value", "unit", "cores", "kind", "gpu_vs_oracle_max_rel_err", "gpu_vs_oracle_sequences"))
        o_cb["sample"] = _cut(cb.get("sample"), 170)
        for k in ("python_all_cores", "c_oracle_all_cores"):
            if isinstance(cb.get(k), dict):
                o_cb[k] = _pick(cb[k], ("value", "cores", "error"))
        out["cpu_baseline"] = o_cb
    if isinstance(res.get("configs"), dict):
        oc = {}
        for name, c in res["configs"].items():
            if not isinstance(c, dict) or "roofline" not in c:
                oc[name] = _pick(c if isinstance(c, dict) else {}, ("skipped", "error"))
                continue
            crf = c["roofline"]
            oc[name] = {**_pick(c, ("tokens_per_s", "ms_per_step", "gpu_vs_oracle_max_rel_err", "status_bits")),
                        **_pick(crf, ("avg_kernel_ms", "frac", "hbm_frac", "algorithmic_frac", "traffic", "traffic_stale")),
                        "kernel_ms": _pick(crf.get("kernel_ms") or {}, ("min", "median", "max")),
                        "frac_bytes": "left L2" if crf.get("traffic") is not None else "compulsory"}
        out["configs"] = oc
    sh = res.get("sharded")
    if isinstance(sh, dict):
        osh = _pick(sh, ("world_sanity", "rows_total", "rows_per_rank", "world_size", "device_count", "backend", "rccl_version",
                         "rccl_high_priority_stream", "build_s", "exchanges_agree", "best_whole_output", "xgmi_peak_GBps", "skipped", "error"))
        if isinstance(sh.get("note"), str):
            osh["note"] = _cut(sh["note"], 120)
        n1 = sh.get("n1_pinned_host")
        if isinstance(n1, dict):
            osh["n1_pinned_host"] = _pick(n1, ("value", "ms_per_step", "pcie_GBps", "pcie_frac", "skipped", "error"))
        z = sh.get("n1_pinned_host_zipf")
        if isinstance(z, dict):
            oz = _pick(z, ("value", "ms_per_step", "different_batch_every_step", "cache_rows", "rows_over_pcie_per_step", "pcie_GBps",
                           "pcie_frac", "status_bits", "prefetch_beats_zero_copy", "skipped", "error"))
            for k in ("zero_copy_same_stream", "zero_copy_static_head_same_hbm"):
                if isinstance(z.get(k), dict):
                    oz[k] = _pick(z[k], ("value", "ms_per_step"))
            so = z.get("scrambled_order")
            if isinstance(so, dict):
                oz["scrambled_order"] = {**_pick(so, ("value", "ms_per_step", "prefetch_beats_static_head")),
                                         "zero_copy_same_stream": (so.get("zero_copy_same_stream") or {}).get("value"),
                                         "zero_copy_static_head_same_hbm": (so.get("zero_copy_static_head_same_hbm") or {}).get("value")}
            osh["n1_pinned_host_zipf"] = oz
        if isinstance(sh.get("exchanges"), dict):
            oe = {}
            for name, e in sh["exchanges"].items():
                if not isinstance(e, dict):
                    continue
                c = _pick(e, ("ms_per_step", "tokens_per_s", "speedup_vs_n1_pinned_host", "status_bits", "scales_with_world", "skipped", "error",
                              "transport_fallback_reason"))
                if isinstance(e.get("with_cu_reserve"), dict):
                    c["with_cu_reserve_ms_per_step"] = e["with_cu_reserve"].get("ms_per_step")
                er = e.get("roofline")
                if isinstance(er, dict):
                    c["hbm_frac"] = er.get("frac")
                    c["xgmi_frac"] = (er.get("wire") or {}).get("frac_of_xgmi_peak")
                if isinstance(e.get("records_transport"), str):
                    c["transport"] = e["records_transport"].split(",")[0].split(" ")[0]
                oe[name] = c
            osh["exchanges"] = oe
        out["sharded"] = osh
    out.update(_pick(res, ("world_sanity", "time_budget_s", "incomplete", "hung_stage", "launcher", "selftest")))
    if details_path:
        out["details"] = details_path
    exact = {k: out[k] for k in ("value", "ms_per_step") if k in out}      # the contract's own figures keep every digit
    out = _sig(out)
    out.update(exact)
    # the limit is a promise: shed the optional blocks, least important first, until the line fits
    for path in (("sharded", "exchanges", "*", "xgmi_frac"), ("sharded", "n1_pinned_host_zipf", "scrambled_order"), ("roofline", "mall_variant"),
                 ("roofline", "hbm_variant"), ("cpu_baseline", "gpu_vs_oracle_sequences"), ("cpu_baseline", "sample"), ("configs",),
                 ("sharded", "n1_pinned_host_zipf"), ("sharded", "exchanges"), ("sharded",), ("config", "workload")):
        if len(json.dumps(out, default=str)) <= LINE_LIMIT:
            break
        node = out
        for k in path[:-1]:
            if k == "*":
                break
            node = node.get(k) if isinstance(node, dict) else None
            if node is None:
                break
        if node is None:
            continue
        if "*" in path:
            for v in node.values():
                if isinstance(v, dict):
                    v.pop(path[-1], None)
        else:
            node.pop(path[-1], None)
        out["line_shortened"] = True
    return out


def details_path_for(n_gpus):
    """Where the whole record goes: $SCONE_BENCH_DETAILS, or gpurun_out/bench_details_n<N>.json under the repo (merged back by
    gpurun), or the temporary directory."""
    p = os.environ.get("SCONE_BENCH_DETAILS")
    if p:
        return p
    d = os.path.join(ROOT, "gpurun_out")
    try:
        os.makedirs(d, exist_ok=True)
        if os.access(d, os.W_OK):
            return os.path.join(d, f"bench_details_n{n_gpus}.json")
    except OSError:
        pass
    import tempfile
    return os.path.join(tempfile.gettempdir(), f"scone_bench_details_n{n_gpus}.json")


class Line:
    """The one JSON line.  `publish` hands over the headline record; from then on every change of it (or of a dict hanging
    off it) is made inside `with line.lock`, and `emit` serialises it inside the same lock -- the watchdog thread can print
    at any moment without meeting a half-built dictionary.  Printed at most once.  Round 5: what is PRINTED is the compact form
    (compact_record: <= LINE_LIMIT characters -- the driver keeps the tail of stdout, and the record had grown to 17 KB); the
    whole record is written to a file named in the line (`details`)."""

    def __init__(self, rank: int) -> None:
        self.lock = threading.RLock()
        self.rank = rank
        self.res = None
        self.headline_done = False        # set on every rank once the timed region and its max-over-ranks are through
        self.emitted = False

    def publish(self, res) -> None:
        with self.lock:
            self.res = res

    def set(self, d, key, value) -> None:
        with self.lock:
            d[key] = value

    def emit(self, incomplete=None) -> bool:
        with self.lock:
            if self.emitted or self.res is None or self.rank != 0:
                return False
            if incomplete:
                self.res["incomplete"] = incomplete
            details = None
            try:                            # the whole record, for whoever wants every phase and provenance string
                details = details_path_for(self.res.get("n_gpus", 1))
                with open(details, "w") as f:
                    json.dump(self.res, f, default=str)
                details = os.path.relpath(details, ROOT) if details.startswith(ROOT + os.sep) else details
            except Exception:
                details = None
            try:
                text = json.dumps(compact_record(self.res, details), default=str)
            except Exception as e:          # never lose the headline to a value json cannot take (or to a bug in the compaction)
                keep = {k: v for k, v in self.res.items() if isinstance(v, (str, int, float, bool, type(None)))}
                keep["incomplete"] = f"{incomplete or ''} (record dropped: {e!r})"
                text = json.dumps(keep)
            try:                            # RCCL prints its banner through C stdio: flush it so the line comes last
                import ctypes
                ctypes.CDLL(None).fflush(None)
            except Exception:
                pass
            sys.stdout.write(text + "\n")
            sys.stdout.flush()
            self.emitted = True
            return True


class Watchdog(threading.Thread):
    """Enforces the job's deadline and the limit of the current stage.  On expiry: rank 0 prints the line (what was
    measured so far, `incomplete` says why), then the process ends through os._exit -- status 0 if the headline was
    measured, 3 if not.  The other ranks follow two seconds later with the same rule.  Nothing is restarted."""

    def __init__(self, budget: Budget, line: Line, rank: int) -> None:
        super().__init__(daemon=True, name="bench-watchdog")
        self.budget, self.line, self.rank = budget, line, rank
        self.stage = None                   # (name, deadline, limit)
        self.grace = 0.0 if rank == 0 else 2.0

    def arm(self, name: str, seconds: float) -> None:
        self.stage = (name, time.time() + seconds, seconds)

    def disarm(self) -> None:
        self.stage = None

    def run(self) -> None:
        while True:
            time.sleep(0.2)
            now = time.time()
            st = self.stage
            if st is not None and now > st[1] + self.grace:
                self.bail(f"stage '{st[0]}' did not complete within its {st[2]:.0f} s; what was measured before it is kept", hung=st[0])
            if now > self.budget.deadline + self.grace:
                self.bail(f"time budget of {self.budget.seconds:.0f} s used up"
                          + (f" in stage '{st[0]}'" if st else "") + "; what was measured until then is kept")

    def bail(self, why: str, hung=None) -> None:
        """Status 0 iff the headline was measured (the line is valid and the driver's launcher must not discard it); a stage
        that HUNG is named in the line (`hung_stage`), and `self_launch` -- this repo's own launcher, used by tests and CI --
        turns that into exit status 4: a hung collective never reads as a clean pass there."""
        code = 3
        try:
            code = 0 if self.line.headline_done else 3
            sys.stderr.write(f"bench.py[rank {self.rank}]: {why}\n")
            sys.stderr.flush()
            if hung is not None:
                with self.line.lock:
                    if self.line.res is not None:
                        self.line.res["hung_stage"] = hung
            self.line.emit(incomplete=why)
        finally:
            os._exit(code)


# ----------------------------------------------------------------------------------------------------------------
# `python bench.py --gpus N` without a launcher: start the N ranks here
def _launch_once(args, n, deadline):
    """One attempt: N fresh children, rank 0's stdout captured.  Returns (rcs, rank-0 stdout, seconds until the first exit)."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    t_start = time.time()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, text=True))
    out_box = {}
    reader = threading.Thread(target=lambda: out_box.setdefault("out", procs[0].stdout.read()), daemon=True)
    reader.start()                                           # drain the pipe while polling, or rank 0 blocks on a full one
    first_exit = None
    while True:
        rcs = [p.poll() for p in procs]
        if first_exit is None and any(rc is not None for rc in rcs):
            first_exit = time.time() - t_start
        if all(rc is not None for rc in rcs):
            break
        # a rank died with an error: the others would sit in their next collective until its own timeout
        if any(rc not in (None, 0) for rc in rcs) or time.time() > deadline:
            time.sleep(3.0)                                  # (ranks that are on their way out through the watchdog)
            for p in procs:
                if p.poll() is None:
                    p.kill()
            rcs = [p.wait() for p in procs]
            break
        time.sleep(0.2)
    reader.join(timeout=10)
    return rcs, out_box.get("out") or "", first_exit or 0.0


def self_launch(args) -> int:
    """Spawn N fresh child processes (one rank per GPU) BEFORE this process touches a GPU -- nothing here calls into HIP,
    and no process that has is ever replaced by another program.  Rank 0's stdout is captured; its last JSON line is
    checked (n_gpus == N) and forwarded as this process's single output line.  The children are polled: as soon as one
    exits non-zero the rest are killed, and nothing outlives the job's time budget (+ 20 s for the ranks' own watchdogs
    to print and leave first).  A line rank 0 did print is forwarded even when a rank failed; the status stays non-zero."""
    n = args.gpus
    one_device = os.environ.get("SCONE_ONE_DEVICE") == "1"
    if not args.selftest:
        import torch                               # device_count() does not initialise the GPU on this image
        have = torch.cuda.device_count()
        if have < n and not one_device:
            print(f"bench.py: --gpus {n} but only {have} HIP device(s) visible", file=sys.stderr)
            return 2
    os.environ.setdefault(T0_ENV, repr(_T_PROCESS_START))
    deadline = float(os.environ[T0_ENV]) + args.time_budget + 20.0
    rcs, out, first_exit = _launch_once(args, n, deadline)
    if rcs[0] not in (0, None) and '{"metric"' not in out and first_exit < 30.0 and time.time() + 60.0 < deadline:
        # the rendezvous port was picked by bind-then-close: another process may have taken it in between.  One retry
        sys.stderr.write(f"bench.py: ranks exited with {rcs} after {first_exit:.0f} s without a result; retrying once on a new port\n")
        rcs, out, first_exit = _launch_once(args, n, deadline)
    line = None
    for ln in out.splitlines():
        if ln.startswith('{"metric"'):
            line = ln
    res = None
    if line is not None:
        try:
            res = json.loads(line)
        except ValueError:
            res = None
    ok = not any(rcs) and res is not None
    if res is not None and res.get("n_gpus") != n:
        sys.stderr.write(f"bench.py: --gpus {n} but the result line says n_gpus = {res.get('n_gpus')}\n")
        return 1
    if res is not None:
        res["launcher"] = f"bench.py started {n} ranks itself (WORLD_SIZE was unset)"
        if not ok:
            res["launcher"] += f"; ranks exited with {rcs}"
        print(json.dumps(res), flush=True)
        if ok and res.get("hung_stage"):                     # the line is valid, the job is not a clean pass
            sys.stderr.write(f"bench.py: stage '{res['hung_stage']}' hung; the line above holds what was measured before it\n")
            return 4
    if not ok:
        sys.stderr.write(f"bench.py: ranks exited with {rcs}; rank 0 printed {'no' if res is None else 'a'} result line\n")
        if out and res is None:
            sys.stderr.write(out[-2000:])
        return 1
    return 0


# ----------------------------------------------------------------------------------------------------------------
def kernel_source_files():
    """The files the timed kernel (k_embed_wave and its siblings) is compiled from: the scone_gather*.hip translation
    units, every header they include (transitively) and the Makefile with the compiler flags."""
    d = os.path.join(ROOT, "scone_amd", "csrc")
    todo = sorted(f for f in os.listdir(d) if f.startswith("scone_gather") and f.endswith(".hip"))
    seen = []
    while todo:
        f = todo.pop(0)
        if f in seen or not os.path.exists(os.path.join(d, f)):
            continue
        seen.append(f)
        for inc in re.findall(r'^\s*#\s*include\s*"([^"]+)"', open(os.path.join(d, f), errors="ignore").read(), flags=re.M):
            todo.append(os.path.normpath(inc))
    return [os.path.join(d, f) for f in sorted(seen)] + [os.path.join(d, "Makefile")]


def _code_only(path: str) -> bytes:
    """The file without comments and without blank space: what the compiler sees.  (Round 5: documentation edits in the
    public header or in a kernel's comments no longer void the committed counter passes; any change of code does.)"""
    text = open(path, errors="ignore").read()
    if os.path.basename(path) == "Makefile":
        text = re.sub(r"(?m)^\s*#.*$", "", text)
    else:
        # string and character literals are kept as they are; // and /* */ comments go
        text = re.sub(r'("(?:\\.|[^"\\\n])*"|\'(?:\\.|[^\'\\\n])*\')|//[^\n]*|/\*.*?\*/',
                      lambda m: m.group(1) or " ", text, flags=re.S)
    return " ".join(text.split()).encode()


def kernel_source_sha() -> str:
    """Hash of the timed kernel's sources (code only, see _code_only): a committed PMC traffic figure is only quoted for the
    code it was measured on."""
    h = hashlib.sha256()
    for p in kernel_source_files():
        if os.path.exists(p):
            h.update(os.path.basename(p).encode())
            h.update(_code_only(p))
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


MAX_DISTINCT_BATCHES = 64       # steps beyond this cycle through the batches: 64 x 0.35 GB of rows is 90x the Infinity Cache


def make_vocabulary(n_rows, keygen, max_n=3, vocab=50257):
    """(vocabulary object for EmbeddingCache.from_synthetic, host keys, host lens) -- host arrays None for `structured`."""
    from scone_amd import NGramExtractor
    from scone_amd import synthetic as S
    if keygen == "structured":
        return S.StructuredVocab(n_rows, vocab=vocab), None, None
    if vocab != S.GPT2_VOCAB:
        raise SystemExit("--vocab needs --keygen structured")
    keys, lens = (S.make_keys if keygen == "zipf" else S.make_keys_torch)(n_rows, S.GPT2_VOCAB, max_n, seed=11)
    return NGramExtractor.from_arrays(keys, lens, max_n=max_n), keys, lens


def make_batches(vocab_obj, keys, lens, stream, B, T, seed, n):
    """`n` DIFFERENT batches of the named stream (same generator, seeds seed, seed + 7919, ...): host arrays of the first one
    (the oracle checks it) and int32 device tensors of all.  S_uniform: f-grams with ids uniform over the table laid end to
    end; S_zipf: iid Zipf(1.1) tokens."""
    import torch
    from scone_amd import synthetic as S
    out, first = [], None
    for i in range(n):
        sd = seed + 7919 * i
        if stream == "uniform":
            t = S.stream_uniform_ids(vocab_obj if keys is None else keys, lens, B, T, sd)
        else:
            t = S.stream_zipf(S.GPT2_VOCAB, B, T, sd)
        if first is None:
            first = t
        out.append(torch.from_numpy(t).to("cuda", torch.int32))
    return first, out


def lookup_loop(cache, batches, wte, wpe, out, steps, warmup, sync, prefetch):
    """The serving loop: step k looks up batches[k % n] and -- `prefetch` -- announces batches[(k + 1) % n] right behind it
    (scone_embed_prefetch, tokens_ready: every batch was generated up front), so that the next step's match runs on the
    handle's side stream beside this step's gather.  W untimed + K timed steps; exactly K matches and K gathers are inside
    the timed region (the first timed batch was announced by the last warm-up step; the last timed step announces a batch
    that is looked up after the region, or never).  Returns measure_lookup's tuple."""
    n = len(batches)
    k = [0]

    def step():
        i = k[0]
        k[0] += 1
        cache.embed_tokens(batches[i % n], wte=wte, wpe=wpe, out=out)
        if prefetch:
            cache.prefetch_tokens(batches[(i + 1) % n], tokens_ready=True)
    return measure_lookup(cache.table, step, batches[0], batches[0].numel(), steps, warmup, sync)


def roofline_block(sig, alg, comp, step_kernel_ms, samples, per_step, n_launch, in_hbm=True, kernel=None):
    """The `roofline` object of one workload.  `frac` = achieved / peak is PHYSICAL: bytes of the launch that crossed the
    L2 <-> fabric boundary (rocprofv3 PMC passes of this kernel source and this workload signature) -- or, without such an
    entry, the compulsory bytes -- over the HIP-event kernel time; it cannot exceed 1.  SURVEY 8d's figure (every row
    REFERENCE counted; cache reuse can carry it past the peak) is `algorithmic_frac`; `hbm_frac` prices the compulsory bytes
    (every distinct row once + output + ids: a lower bound on what HBM moves)."""
    tr, stale = read_traffic(sig)
    traffic = None if (tr is None or stale) else tr.get("hbm_bytes_per_launch")
    per_s = step_kernel_ms * 1e-3
    if traffic is not None:
        phys_bytes, phys_kind = traffic, ("bytes that left L2 per launch (2 x FETCH_SIZE + WRITE_SIZE, rocprofv3 PMC passes of this "
                                          "kernel source: an upper bound on HBM bytes, Infinity-Cache hits included)")
    else:
        phys_bytes, phys_kind = comp, ("compulsory bytes per launch (every distinct table row and wte row once + output + "
                                       "ids: a lower bound on HBM bytes; no PMC entry for this workload and kernel source)")
    achieved = phys_bytes / per_s / 1e9
    return {
        "bound": "hbm",
        "limited_by": None if in_hbm else "PCIe Gen5 x16 (~63 GB/s): the table's rows live in pinned host DRAM",
        "kernel": kernel or "scone_gather::k_embed_wave (gather+dequant+reduce+combine), HIP-event timed",
        "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBPS,
        "frac_kind": phys_kind + f" / avg_kernel_ms ({step_kernel_ms:.4f} ms, HIP events) / 8 TB/s",
        "frac_bytes": phys_bytes,
        "algorithmic_bytes_per_launch": alg, "algorithmic_GBps": alg / per_s / 1e9,
        "algorithmic_frac": alg / per_s / 1e9 / HBM_PEAK_GBPS,
        "avg_kernel_ms": step_kernel_ms,
        "kernel_ms": kernel_stats(samples, per_step), "timed_launches": n_launch, "launches_per_step": per_step,
        "hbm_bytes_compulsory": comp if in_hbm else None,
        "hbm_frac": comp / per_s / 1e9 / HBM_PEAK_GBPS if in_hbm else None,
        "traffic": traffic,
        "traffic_source": None if tr is None else tr.get("source"),
        "traffic_stale": bool(stale),
        "traffic_GBps": None if traffic is None else traffic / per_s / 1e9,
        "traffic_frac": None if traffic is None else traffic / per_s / 1e9 / HBM_PEAK_GBPS,
        "kernel_source_sha": kernel_source_sha(),
    }


def workload_sig(fmt, d, N, B, T, stream, placement="hbm", keygen="zipf", rotated=True, extra="", vocab=50257):
    return (f"{fmt}-d{d}-N{N}-B{B}-T{T}-{stream}-{placement}" + extra + (f"-V{vocab}" if vocab != 50257 else "")
            + {"zipf": "", "zipf_gpu": "-zipfgpu", "structured": "-structured"}[keygen] + ("-rot" if rotated else ""))


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


def config_record(name, fmt, d, N, keygen, stream, B, T, steps, warmup, sync, prefetch, vocab_cache=None, wte=None, wpe=None,
                  check=True, vocab=50257):
    """One single-GPU workload measured like the headline: its own table, a different batch every step, the serving loop with
    the next batch announced, HIP-event kernel times (min / median / max), counter-priced `frac` when profiles/hbm_traffic.json
    holds passes for this signature and kernel source, and the GPU output of 8 sequences of the first batch checked against
    the oracle.  `vocab_cache`: (vocabulary, keys, lens) to re-use (the headline's 1M-row vocabulary serves C2)."""
    import torch
    from scone_amd import EmbeddingCache
    from scone_amd import synthetic as S
    from scone_amd.hip_backend import format_code
    seed, base_scale = 7, 0.02 / 127
    t_build = time.perf_counter()
    vocab_obj, keys, lens = vocab_cache if vocab_cache is not None else make_vocabulary(N, keygen, vocab=vocab)
    kw = {"n_rows": N} if keys is None else {}
    cache = EmbeddingCache.from_synthetic(vocab_obj, d, table_format=fmt, seed=seed, base_scale=base_scale, **kw)
    if wte is None:
        g = torch.Generator(device="cuda").manual_seed(5)
        wte = (torch.randn(vocab, d, generator=g, device="cuda") * 0.02).half()
        wpe = (torch.randn(1024, d, generator=g, device="cuda") * 0.01).half()
    n_b = min(steps + warmup, MAX_DISTINCT_BATCHES)
    tok_np, batches = make_batches(vocab_obj, keys, lens, stream, B, T, 1234, n_b)
    out = torch.empty(B, T, d, dtype=torch.float16, device="cuda")
    table = cache.table
    torch.cuda.synchronize()
    t_build = time.perf_counter() - t_build
    alg, comp, sum_k, k_hist, nr, nt = workload_bytes(table, batches[0], format_code(fmt), d)
    dt, n_launch, kern_ms, samples = lookup_loop(cache, batches, wte, wpe, out, steps, warmup, sync, prefetch)
    avg_ms = kern_ms / max(n_launch, 1)
    sig = workload_sig(fmt, d, N, B, T, stream, "hbm", keygen, rotated=n_b > 1, vocab=vocab)
    rf = roofline_block(sig, alg, comp, avg_ms, samples, 1, n_launch)
    res = {
        "name": name,
        "workload": f"{N}-row {fmt} f-gram table d={d} max_n=3 in HBM ({keygen} vocabulary), S_{stream} stream, {B}x{T} tokens/step, "
                    f"a different batch every step ({n_b} batches); fused match+gather+dequant+mean+wte+wpe, fp16 out; "
                    f"{nr} distinct table rows and {nt} distinct wte rows in the first batch",
        "workload_sig": sig, "tokens_per_s": B * T * steps / dt, "ms_per_step": dt / steps * 1e3, "steps": steps, "warmup": warmup,
        "step_minus_kernel_us": (dt / steps * 1e3 - avg_ms) * 1e3, "next_batch_announced": bool(prefetch),
        "mean_hits_per_token": sum_k / (B * T), "hits_histogram_K0_6": k_hist[:7], "build_s": t_build,
        "roofline": rf, "status_bits": int(table.status()),
    }
    if check:
        try:
            cache.embed_tokens(batches[0], wte=wte, wpe=wpe, out=out)
            torch.cuda.synchronize()
            err, picks = cpu_baseline_spot_check(N, keys, lens, tok_np, out, fmt, d, seed, base_scale, wte, wpe, vocab=vocab)
            res["gpu_vs_oracle_max_rel_err"], res["gpu_vs_oracle_sequences"] = err, picks
        except Exception as e:
            res["gpu_vs_oracle_max_rel_err"], res["gpu_vs_oracle_error"] = None, repr(e)
    del cache, table, batches, out
    torch.cuda.empty_cache()
    return res


def hbm_variant(args, wte, wpe, sync, prefetch=True):
    """The headline's format and dim on a workload that defeats the caches: 10M rows (7.7 GB of INT8 d = 768 rows -- 30x
    the Infinity Cache), structured vocabulary (token ids uniform over the 50,257-word vocabulary, one bigram / trigram
    row per window, each referenced by the 2-3 adjacent tokens it covers and by nothing else in the launch), a different
    batch every step."""
    steps = max(10, min(args.steps, 30))
    r = config_record("hbm_variant", args.format, args.dim, 10_000_000, "structured", "uniform", args.batch, args.seq, steps, 3,
                      sync, prefetch, wte=wte if args.dim == wte.shape[1] else None, wpe=wpe if args.dim == wpe.shape[1] else None,
                      check=True)
    rf = r["roofline"]
    return {"workload": r["workload"], "workload_sig": r["workload_sig"], "mean_hits_per_token": r["mean_hits_per_token"],
            "avg_kernel_ms": rf["avg_kernel_ms"], "kernel_ms": rf["kernel_ms"], "tokens_per_s": r["tokens_per_s"],
            "ms_per_step": r["ms_per_step"],
            "algorithmic_bytes_per_launch": rf["algorithmic_bytes_per_launch"], "algorithmic_GBps": rf["algorithmic_GBps"],
            "algorithmic_frac": rf["algorithmic_frac"], "hbm_bytes_compulsory": rf["hbm_bytes_compulsory"],
            "hbm_GBps": rf["hbm_bytes_compulsory"] / rf["avg_kernel_ms"] / 1e6, "hbm_frac": rf["hbm_frac"],
            "traffic": rf["traffic"], "traffic_frac": rf["traffic_frac"], "traffic_stale": rf["traffic_stale"],
            "gpu_vs_oracle_max_rel_err": r.get("gpu_vs_oracle_max_rel_err"), "status_bits": r.get("status_bits")}


def _host_memory_available():
    import psutil
    avail = psutil.virtual_memory().available
    for f_lim, f_use in (("/sys/fs/cgroup/memory.max", "/sys/fs/cgroup/memory.current"),
                         ("/sys/fs/cgroup/memory/memory.limit_in_bytes", "/sys/fs/cgroup/memory/memory.usage_in_bytes")):
        try:                                            # a container's own limit counts, not only the host's free memory
            lim = open(f_lim).read().strip()
            if lim != "max":
                avail = min(avail, int(lim) - int(open(f_use).read().strip()))
        except (OSError, ValueError):
            pass
    return avail


def pinned_baseline(args, sync, zipf_too=True):
    """How ONE GPU serves a table that does not fit its HBM -- rows in pinned host DRAM (BASELINE config C4: 100M rows
    INT4 d = 1024 = 52.8 GB, first 1M rows hot in HBM), on the batch shape of the sharded record.  PCIe-bound, so the
    rate barely depends on the table's size; this is what ">= 4x at 8 GPUs vs 1 GPU on the 1B-row sharded table" is
    computed against.  Returns (record for the S_uniform stream with the rows read in place over PCIe -- the faster
    mechanism on that stream --, record for the Zipf-ids stream: f-gram ids drawn from a power law over the
    frequency-ordered table, what real text looks like to such a table; None unless `zipf_too`).

    Round 4, the Zipf record: a DIFFERENT batch every step (round 3 re-used one batch, which says nothing about anything that
    keeps rows between steps).  `value` = the north-star's "async prefetch" as it is built now -- a persistent HBM cache of
    cold rows (clock eviction) in front of the chunk pipeline, warmed by `warmup_batches` steps of the same stream --, with
    the rows that crossed PCIe per step from the library's counters; beside it, on the same batches: the rows read in place
    (`zero_copy_same_stream`), and -- the honest alternative for the same HBM -- the static hot head enlarged by the cache's
    rows (`zero_copy_static_head_same_hbm`: ids ARE frequency-ordered, so on a stationary stream no cache can beat it; the
    cache is for traffic that drifts away from the order the table was built in)."""
    import torch
    from scone_amd import EmbeddingCache
    from scone_amd import synthetic as S
    N, d, B, T = args.pinned_rows, 1024, 2048, 512
    need = N * 512 + 8e9
    avail = _host_memory_available()
    if avail < need:
        skip = {"value": None, "skipped": f"needs {need / 1e9:.0f} GB of host memory for the pinned table ({avail / 1e9:.0f} GB available)"}
        return skip, (dict(skip) if zipf_too else None)
    hot = min(1_000_000, max(N // 100, 1))
    vocab = S.StructuredVocab(N)
    g = torch.Generator(device="cuda").manual_seed(5)
    wte = (torch.randn(S.GPT2_VOCAB, d, generator=g, device="cuda") * 0.02).half()
    wpe = (torch.randn(1024, d, generator=g, device="cuda") * 0.01).half()
    out = torch.empty(B, T, d, dtype=torch.float16, device="cuda")

    def run(cache, toks, steps=4, warm=None, prefetch=False):
        """ms per step over `steps` lookups of toks[i % len(toks)], after one untimed lookup of every batch in `warm`
        (default: the first batch; [] = none -- a cache must not have seen the timed batches).  prefetch: the loop of a server
        that knows its next tokens early -- scone_embed_prefetch of batch i + 1 is issued right after the lookup of batch i is
        queued (the tokens were generated up front: tokens_ready), so the next batch's first chunks are matched, placed and
        copied beside this batch's last lookups."""
        cache.table.reserve(B * T)
        for t in (toks[:1] if warm is None else warm):
            cache.embed_tokens(t, wte=wte, wpe=wpe, out=out)
        sync()
        t0 = time.perf_counter()
        for i in range(steps):
            cache.embed_tokens(toks[i % len(toks)], wte=wte, wpe=wpe, out=out)
            if prefetch and i + 1 < steps:
                cache.prefetch_tokens(toks[(i + 1) % len(toks)], tokens_ready=True)
        sync()
        return (time.perf_counter() - t0) / steps

    def table(**kw):
        return EmbeddingCache.from_synthetic(vocab, d, table_format="int4", seed=7, base_scale=0.02 / 127, n_rows=N,
                                             placement="pinned_host", **kw)

    cache = table(hot_rows=hot)
    tok = torch.from_numpy(S.stream_uniform_ids(vocab, None, B, T, 1234)).to("cuda", torch.int32)
    dt = run(cache, [tok])
    _, ids_u = cache.table.match_csr(tok)
    cold_u = ids_u[ids_u >= hot]
    n_cold_ref, n_cold_distinct = int(cold_u.numel()), int(torch.unique(cold_u).numel())
    del ids_u, cold_u
    res = {"value": B * T / dt, "unit": "tokens/s", "ms_per_step": dt * 1e3, "steps": 4,
           "workload": f"{N}-row int4 table d={d} in pinned host DRAM (rows read in place over PCIe), first {hot} rows in HBM, "
                       f"structured vocabulary, S_uniform, {B}x{T} tokens/step",
           "bound": "PCIe Gen5 x16 (~64 GB/s)", "pcie_peak_GBps": PCIE_PEAK_GBPS,
           "cold_row_references": n_cold_ref, "distinct_cold_rows": n_cold_distinct,
           "bytes_over_pcie_per_step_at_least": n_cold_distinct * 528,
           "pcie_GBps": n_cold_distinct * 528 / dt / 1e9, "pcie_frac": n_cold_distinct * 528 / dt / 1e9 / PCIE_PEAK_GBPS,
           "pcie_frac_kind": "every DISTINCT cold row of the batch once (528 B payload; a row re-referenced after it left L2 crosses "
                             "again, so this is a lower bound on the link's bytes) / ms_per_step / 64 GB/s"}
    zres = None
    if zipf_too:
        try:
            steps, warm_n = args.pinned_zipf_steps, args.pinned_zipf_warmup
            cache_rows = min(args.pinned_cache_rows, max(N - hot, 1))
            timed = [S.stream_zipf_ids_torch(vocab, B, T, 50_000 + i) for i in range(steps)]     # never seen by any warm-up
            st = []
            for t in timed[:2]:
                _, ids = cache.table.match_csr(t)
                cold = ids[ids >= hot]
                st.append((float(ids.numel()) / (B * T), int(cold.numel()), int(torch.unique(cold).numel())))
                del ids, cold
            stats = {"mean_hits_per_token": sum(x[0] for x in st) / len(st), "cold_row_references": sum(x[1] for x in st) / len(st),
                     "distinct_cold_rows": sum(x[2] for x in st) / len(st)}
            dt_zero = run(cache, timed, steps)
            # the same law over a table whose order is NOT the traffic's frequency order (synthetic.stream_zipf_ids_torch,
            # scramble=True): what the cache is for -- the static head holds 1 % / 17 % of the rows, not of the references
            timed_s = [S.stream_zipf_ids_torch(vocab, B, T, 60_000 + i, scramble=True) for i in range(steps)]
            _, ids = cache.table.match_csr(timed_s[0])
            cold = ids[ids >= hot]
            stats_s = {"mean_hits_per_token": float(ids.numel()) / (B * T), "cold_row_references": int(cold.numel()),
                       "distinct_cold_rows": int(torch.unique(cold).numel())}
            del ids, cold
            dt_zero_s = run(cache, timed_s, steps)
            del cache
            torch.cuda.empty_cache()
            # the same table behind the persistent cache of cold rows (stage_tokens / cache_rows are properties of the handle)
            cache = table(hot_rows=hot, stage_tokens=args.pinned_stage_tokens, cache_rows=cache_rows)
            warm = [S.stream_zipf_ids_torch(vocab, B, T, 1234 + i) for i in range(warm_n)]
            cache.table.reserve(B * T)
            for t in warm:
                cache.embed_tokens(t, wte=wte, wpe=wpe, out=out)
            del warm
            sync()
            c0 = cache.table.stage_counters()
            dt_cached = run(cache, timed, steps, warm=[], prefetch=True)
            c1 = cache.table.stage_counters()
            copied = (c1["rows_copied"] - c0["rows_copied"]) / steps
            status = cache.table.status()
            for i in range(warm_n):                 # the cache re-learns the scrambled stream (its rows are elsewhere)
                cache.embed_tokens(S.stream_zipf_ids_torch(vocab, B, T, 70_000 + i, scramble=True), wte=wte, wpe=wpe, out=out)
            sync()
            c2 = cache.table.stage_counters()
            dt_cached_s = run(cache, timed_s, steps, warm=[], prefetch=True)
            copied_s = (cache.table.stage_counters()["rows_copied"] - c2["rows_copied"]) / steps
            status |= cache.table.status()
            del cache
            torch.cuda.empty_cache()
            cache = table(hot_rows=hot + cache_rows)
            dt_static = run(cache, timed, steps)
            dt_static_s = run(cache, timed_s, steps)
            zres = {"value": B * T / dt_cached, "unit": "tokens/s", "ms_per_step": dt_cached * 1e3, "steps": steps,
                    "different_batch_every_step": True, "warmup_batches": warm_n,
                    "mechanism": f"persistent HBM cache of cold rows ({c1['cache_rows']} row slots = {c1['cache_rows'] * 528 / 1e9:.1f} GB, "
                                 f"clock eviction) in front of the chunk pipeline ({c1['chunk_tokens']}-token chunks: match, touch / "
                                 "place, remap and the copy of the missing rows host -> HBM on side streams while the previous "
                                 "chunk is reduced); scone_embed_prefetch of batch i + 1 issued right after the lookup of batch i",
                    "cache_rows": c1["cache_rows"], "stage_tokens": c1["chunk_tokens"],
                    "rows_over_pcie_per_step": copied, "bytes_over_pcie_per_step": copied * 528,
                    "pcie_peak_GBps": PCIE_PEAK_GBPS, "pcie_GBps": copied * 528 / dt_cached / 1e9,
                    "pcie_frac": copied * 528 / dt_cached / 1e9 / PCIE_PEAK_GBPS,
                    "pcie_frac_kind": "rows copied host -> HBM per step (the library's counter) x 528 B / ms_per_step / 64 GB/s: the link "
                                      "is NOT the bound of the cached step -- the lookup out of [hot head | cache] is (HBM), which is "
                                      "the point of the cache",
                    "cache_hit_rate_of_distinct_cold_rows": 1.0 - copied / max(stats["distinct_cold_rows"], 1.0),
                    "status_bits": status,
                    "zero_copy_same_stream": {"value": B * T / dt_zero, "ms_per_step": dt_zero * 1e3,
                                              "bytes_over_pcie_per_step_at_least": stats["distinct_cold_rows"] * 528},
                    "zero_copy_static_head_same_hbm": {"value": B * T / dt_static, "ms_per_step": dt_static * 1e3,
                                                       "hot_rows": hot + cache_rows},
                    "prefetch_beats_zero_copy": bool(dt_cached <= dt_zero),
                    "scrambled_order": {
                        "what": "the same power law, popularity rank r served by row (r * 61803399) % N: the table's order is not "
                                "the traffic's frequency order (built on one corpus, served on another); same table, same cache "
                                f"(re-warmed by {warm_n} batches of this stream), same three mechanisms",
                        "value": B * T / dt_cached_s, "ms_per_step": dt_cached_s * 1e3, "rows_over_pcie_per_step": copied_s,
                        "zero_copy_same_stream": {"value": B * T / dt_zero_s, "ms_per_step": dt_zero_s * 1e3},
                        "zero_copy_static_head_same_hbm": {"value": B * T / dt_static_s, "ms_per_step": dt_static_s * 1e3},
                        "prefetch_beats_zero_copy": bool(dt_cached_s <= dt_zero_s),
                        "prefetch_beats_static_head": bool(dt_cached_s <= dt_static_s), **stats_s},
                    "workload": f"{N}-row int4 table d={d} in pinned host DRAM, first {hot} rows in HBM, structured vocabulary, "
                                f"S_zipf_ids (f-grams laid end to end, ids ~ bounded power law with exponent 1.1 over the "
                                f"frequency-ordered table: the realistic stream), {B}x{T} tokens/step, a different batch every step", **stats,
                    "bound": "PCIe Gen5 x16 (~64 GB/s)"}
        except Exception as e:
            zres = {"value": None, "error": repr(e)}
    del cache, tok, out, wte, wpe
    torch.cuda.empty_cache()
    return res, zres


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


# ----------------------------------------------------------------------------------------------------------------
def selftest_main(args):
    """CPU rehearsal of the time budget (tests/test_bench_budget.py): no GPU, gloo, a made-up headline clearly marked as
    such, then two stages of `run_stages` -- the second one a collective that rank 1 never joins when `--selftest hang`.
    Everything that matters is the real code: Budget, Line, Watchdog, run_stages, self_launch, the exit path."""
    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    budget = Budget(args.time_budget)
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
        sys.exit(self_launch(args))
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
    budget = Budget(args.time_budget)
    line = Line(rank)
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
    # ---- the cache-defeating variant, the CPU baselines, the sharded record: outside the timed region -----------
    if rank == 0 and world == 1 and not sharded and emu is None and not args.no_hbm_variant and args.placement == "hbm" \
            and budget.remaining() > 150.0:
        watchdog.arm("roofline.hbm_variant", min(150.0, budget.remaining() - 60.0))
        try:
            hv = hbm_variant(args, wte, wpe, sync, prefetch)
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
                mv = config_record("mall_variant", args.format, args.dim, 10_000_000, "structured", "uniform", args.batch, args.seq,
                                   max(10, min(args.steps, 30)), 3, sync, prefetch, check=True, vocab=262144)
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
                                  prefetch, vocab_cache=reuse, wte=wte if cd == d else None, wpe=wpe if cd == d else None)
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


if __name__ == "__main__":
    main()
