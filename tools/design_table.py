#!/usr/bin/env python3
"""Markdown rows of DESIGN.md section 5's table from a committed configs.jsonl (tools/run_configs.sh):  tools/design_table.py profiles/r06z"""
import json, sys
d = sys.argv[1]
for ln in open(d + "/configs.jsonl"):
    if not ln.strip():
        continue
    c = json.loads(ln)
    rf = c["roofline"]
    name = c["config_name"]
    if rf.get("traffic") is not None:
        print("| %s | %.3f G | %.3f ms | %.3f ms | %.2f GB | %.3f … %.3f (%.3f) | %.2f | %.2f |" % (
            name, c["value"] / 1e9, c["ms_per_step"], rf["avg_kernel_ms"], rf["traffic"] / 1e9, rf["frac_lo"], rf["frac"], rf["frac_profile_box"],
            rf["hbm_frac"], rf["algorithmic_frac"]))
    else:
        print("| %s | %.3f G | %.3f ms | -- | -- | -- | -- | -- |" % (name, c["value"] / 1e9, c["ms_per_step"]))
