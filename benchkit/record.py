"""The printed line of bench.py: the time budget, the compact form of the record (<= LINE_LIMIT characters), the watchdog that
prints what has been measured when a stage hangs.  No GPU, no torch: importable anywhere (tests/test_host_logic.py,
tests/test_bench_budget.py)."""
import json
import os
import sys
import threading
import time

from .common import ROOT, T0_ENV


class Budget:
    """One absolute deadline for the whole job (wall clock, shared with the ranks `self_launch` starts)."""

    def __init__(self, seconds: float, process_start=None) -> None:
        self.t0 = float(os.environ.get(T0_ENV) or process_start or time.time())
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
    if isinstance(cfg.get("output_buffer"), dict):
        out["config"]["output_buffer"] = _sig(_pick(cfg["output_buffer"], ("candidates", "kernel_ms", "finalists_kernel_ms", "kept")), 4)
    out["config"]["workload"] = _cut(cfg.get("workload"), 160)
    out["config"]["parallelism"] = _cut(cfg.get("parallelism"), 110)
    left_l2 = rf.get("traffic") is not None

    def roof(r, extra=()):
        # (not repeated in the line: frac_bytes = traffic or the compulsory bytes, traffic_frac = frac when traffic is quoted,
        # timed_launches = kernel_ms.n; the details file has them)
        c = _pick(r, ("bound", "limited_by", "achieved", "peak", "unit", "frac", "algorithmic_bytes_per_launch",
                      "algorithmic_frac", "avg_kernel_ms", "kernel_ms", "hbm_bytes_compulsory", "hbm_frac", "traffic",
                      "traffic_stale", "kernel_source_sha", "frac_lo", "frac_hi", "frac_profile_box",
                      "profile_kernel_ms") + tuple(extra))
        if "frac_kind" in r:
            c["frac_kind"] = ("left L2: 2*FETCH_SIZE+WRITE_SIZE (rocprofv3 PMC, this kernel source)" if r.get("traffic") is not None
                              else "compulsory bytes (no PMC entry)") + " / HIP-event kernel time / 8 TB/s"
        if r.get("traffic_source"):
            c["traffic_source"] = r["traffic_source"].split(":")[0]
        return c
    o_rf = roof(rf, ("match_us",))
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
        o_cb["sample"] = _cut(cb.get("sample"), 120)
        for k in ("python_all_cores", "c_oracle_all_cores"):
            if isinstance(cb.get(k), dict):
                o_cb[k] = _pick(cb[k], ("value", "cores", "error"))
        out["cpu_baseline"] = o_cb
    if isinstance(res.get("configs"), dict):
        oc = {}
        for name, c in res["configs"].items():
            if not isinstance(c, dict) or "roofline" not in c:
                # (C1 has no roofline block: a 4096-token call is launch-bound; its figures are the call rate and the oracle check)
                oc[name] = _pick(c if isinstance(c, dict) else {}, ("skipped", "error", "tokens_per_s", "ms_per_step", "gpu_vs_oracle_bit_exact",
                                                                    "gpu_vs_oracle_max_rel_err", "cpu_port_1core_tokens_per_s",
                                                                    "get_token_embeddings_ms_per_512_token_sequence", "status_bits"))
                continue
            crf = c["roofline"]
            oc[name] = {**_pick(c, ("tokens_per_s", "ms_per_step", "gpu_vs_oracle_max_rel_err", "status_bits")),
                        **({"wall_outlier": True} if c.get("wall_outlier") else {}),
                        **_pick(crf, ("avg_kernel_ms", "frac", "frac_lo", "frac_profile_box", "hbm_frac", "algorithmic_frac", "traffic")),
                        "kernel_ms": _pick(crf.get("kernel_ms") or {}, ("min", "median", "max"))}
            if crf.get("traffic") is None:          # said only when it is NOT the counter-priced case
                oc[name]["frac_bytes"] = "compulsory" + (" (traffic entry stale)" if crf.get("traffic_stale") else "")
        out["configs"] = oc
    lat = res.get("latency")
    if isinstance(lat, dict):
        out["latency"] = {k: (_sig(_pick(v, ("call_us", "sync_us", "kernel_us", "graph_us", "form", "graph_error")), 3) if isinstance(v, dict) else v)
                          for k, v in lat.items()}
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
        if isinstance(sh.get("exchanges"), dict) and sh["exchanges"]:
            # N > 1: what the first 8-GPU contact is read for, in a block of its own that the shedding below never touches
            dp = sh["exchanges"].get("rows_slices_only") if isinstance(sh["exchanges"].get("rows_slices_only"), dict) else {}
            out["sharded_summary"] = {
                **_pick(sh, ("world_size", "rccl_version", "exchanges_agree", "rows_total")),
                "best_whole_output": _pick(sh.get("best_whole_output") or {}, ("exchange", "tokens_per_s", "ms_per_step",
                                                                               "speedup_vs_n1_pinned_host")),
                "form_for_data_parallel_consumers": {"exchange": "rows_slices_only",
                                                     **_pick(dp, ("tokens_per_s", "ms_per_step", "speedup_vs_n1_pinned_host",
                                                                  "status_bits", "error", "skipped"))},
                "n1_pinned_host_tokens_per_s": (sh.get("n1_pinned_host") or {}).get("value"),
                "hung_stage": res.get("hung_stage")}
    out.update(_pick(res, ("world_sanity", "time_budget_s", "incomplete", "hung_stage", "launcher", "selftest")))
    if details_path:
        out["details"] = details_path
    exact = {k: out[k] for k in ("value", "ms_per_step") if k in out}      # the contract's own figures keep every digit
    out = _sig(out)
    out.update(exact)
    # the limit is a promise: shed the optional blocks, least important first, until the line fits
    for path in (("sharded", "exchanges", "*", "xgmi_frac"), ("sharded", "n1_pinned_host_zipf", "scrambled_order"), ("roofline", "mall_variant"),
                 ("roofline", "hbm_variant"), ("cpu_baseline", "gpu_vs_oracle_sequences"), ("cpu_baseline", "sample"), ("latency",), ("configs",),
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
