#!/usr/bin/env python3
"""Short human-readable digest of one bench.py line (last JSON line of the file given)."""
import json
import sys

text = open(sys.argv[1]).read().strip()
try:
    r = json.loads(text)                       # a committed, indented record
except ValueError:
    r = json.loads(text.splitlines()[-1])      # a captured stdout: the line is the last one
rf = r["roofline"]
print("headline  %.3f G tok/s  step %.4f ms  kernel %.4f ms %s  frac %.3f (%s)  alg %.3f  hbm %.3f  match_us %s  batches %s" % (
    r["value"] / 1e9, r["ms_per_step"], rf["avg_kernel_ms"], rf.get("kernel_ms"), rf["frac"],
    "left L2" if rf.get("traffic") is not None else "compulsory", rf["algorithmic_frac"], rf.get("hbm_frac") or 0,
    rf.get("match_us"), r["config"].get("distinct_batches")))
if rf.get("same_batch"):
    sb = rf["same_batch"]
    print("same_batch  step %.4f ms  kernel %.4f ms" % (sb["ms_per_step"], sb["avg_kernel_ms"]))
hv = rf.get("hbm_variant")
if hv:
    print("hbm_variant", {k: hv.get(k) for k in ("tokens_per_s", "avg_kernel_ms", "hbm_frac", "traffic_frac", "error")})
mv = rf.get("mall_variant")
if mv:
    print("mall_variant", {k: mv.get(k) for k in ("tokens_per_s", "ms_per_step", "gpu_vs_oracle_max_rel_err", "error")},
          {k: mv.get("roofline", mv).get(k) for k in ("avg_kernel_ms", "hbm_frac", "traffic_frac", "algorithmic_frac")})
for k, c in (r.get("configs") or {}).items():
    crf = c.get("roofline", c)          # (the printed line holds the config's figures flat; the details file nests them)
    print("config", k, {x: c.get(x) for x in ("tokens_per_s", "ms_per_step", "build_s", "gpu_vs_oracle_max_rel_err", "status_bits", "skipped", "error") if c.get(x) is not None},
          crf.get("kernel_ms"), "frac", crf.get("frac"), "stale", crf.get("traffic_stale"))
cb = r.get("cpu_baseline")
if cb:
    print("cpu", {k: cb.get(k) for k in ("value", "cores", "gpu_vs_oracle_max_rel_err")})
s = r.get("sharded") or {}
for k in ("n1_pinned_host", "n1_pinned_host_zipf"):
    if isinstance(s.get(k), dict):
        print(k, {x: s[k].get(x) for x in ("value", "ms_per_step", "pcie_GBps", "pcie_frac", "error", "skipped") if s[k].get(x) is not None})
for k in ("incomplete", "hung_stage"):
    if r.get(k):
        print(k, r[k])
