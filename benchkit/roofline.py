"""Pricing a measured kernel time against the HBM roofline: the workload's byte counts, the committed counter passes
(profiles/hbm_traffic.json, quoted only for the kernel source they were taken on), the `roofline` block of the line."""
import hashlib
import json
import os
import re

from .common import HBM_PEAK_GBPS, ROOT


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


def roofline_block(sig, alg, comp, step_kernel_ms, samples, per_step, n_launch, in_hbm=True, kernel=None):
    """The `roofline` object of one workload.  `frac` = achieved / peak is PHYSICAL: bytes of the launch that crossed the
    L2 <-> fabric boundary (rocprofv3 PMC passes of this kernel source and this workload signature) -- or, without such an
    entry, the compulsory bytes -- over the HIP-event kernel time; it cannot exceed 1.  SURVEY 8d's figure (every row
    REFERENCE counted; cache reuse can carry it past the peak) is `algorithmic_frac`; `hbm_frac` prices the compulsory bytes
    (every distinct row once + output + ids: a lower bound on what HBM moves)."""
    tr, stale = read_traffic(sig)
    traffic = None if (tr is None or stale) else tr.get("hbm_bytes_per_launch")
    per_s = step_kernel_ms * 1e-3
    rng = frac_range(None if (tr is None or stale) else tr, step_kernel_ms)
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
        **rng,
    }


READ_FACTOR_LO = 1.74   # profiles/r01f/fetch_size_calibration.json: launches of INT8 row loads alone (8 B and 4 B per lane)
READ_FACTOR_HI = 2.00   # the guide's gfx950 correction, confirmed by the same calibration for 16-B-per-lane fp16 row loads


def frac_range(entry, kernel_ms):
    """How wide `frac` is, and what it is on the box the counters came from.  FETCH_SIZE under-reports reads by a factor that
    depends on the access width: 2.00 for 16-B-per-lane loads (the guide; the wte rows here), 1.74 measured for launches
    dominated by INT8 row loads -- a real launch mixes the two, so the bytes that left L2 lie between
    1.74 x FETCH_SIZE + WRITE_SIZE (`frac_lo`) and 2.00 x FETCH_SIZE + WRITE_SIZE (`frac_hi`; `frac` itself quotes this upper
    end).  `frac_profile_box`: the upper-end bytes over the kernel time rocprofv3 recorded in the SAME profile
    (profiles/<tag>/kernel_stats.csv) -- reproducible from profiles/ alone; `frac` divides by THIS run's HIP-event time."""
    if not entry or entry.get("fetch_size_kb_raw") is None or entry.get("write_size_kb") is None:
        return {}
    f, w = entry["fetch_size_kb_raw"] * 1024.0, entry["write_size_kb"] * 1024.0
    per_s = kernel_ms * 1e-3
    out = {"frac_lo": (READ_FACTOR_LO * f + w) / per_s / 1e9 / HBM_PEAK_GBPS,
           "frac_hi": (READ_FACTOR_HI * f + w) / per_s / 1e9 / HBM_PEAK_GBPS,
           "frac_read_factor": [READ_FACTOR_LO, READ_FACTOR_HI]}
    pk = entry.get("profile_kernel_ms")
    if pk:
        out["profile_kernel_ms"] = pk
        out["frac_profile_box"] = (READ_FACTOR_HI * f + w) / (pk * 1e-3) / 1e9 / HBM_PEAK_GBPS
    return out


def workload_sig(fmt, d, N, B, T, stream, placement="hbm", keygen="zipf", rotated=True, extra="", vocab=50257):
    return (f"{fmt}-d{d}-N{N}-B{B}-T{T}-{stream}-{placement}" + extra + (f"-V{vocab}" if vocab != 50257 else "")
            + {"zipf": "", "zipf_gpu": "-zipfgpu", "structured": "-structured"}[keygen] + ("-rot" if rotated else ""))
