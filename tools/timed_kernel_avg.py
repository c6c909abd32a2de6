#!/usr/bin/env python3
"""The rocprofv3 kernel trace of a bench run -> the average duration of the LAST `steps` dispatches of the gather kernel (the timed
steps: the warm-up launches and, since round 6, the trial lookups of `alloc_output` into the candidate output buffers come before
them).  This -- not the all-calls average of kernel_stats.csv -- is what must agree with the HIP-event figure of the line.
    tools/timed_kernel_avg.py <dir with *kernel_trace.csv> <steps> <out.json>"""
import csv
import glob
import json
import sys


def main(d, steps, out):
    f = glob.glob(d + "/*/*kernel_trace.csv") + glob.glob(d + "/*kernel_trace.csv")
    rows = []
    for r in csv.DictReader(open(f[0])):
        if "k_embed_wave" in r["Kernel_Name"] or "k_embed<" in r["Kernel_Name"]:
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), r["Kernel_Name"].split("(")[0]))
    rows.sort()
    last = rows[-int(steps):]
    durs = [x[1] for x in last]
    json.dump({"kernel": last[-1][2][-60:], "timed_launches": len(durs), "avg_ns": sum(durs) / len(durs), "min_ns": min(durs),
               "max_ns": max(durs), "all_calls": len(rows), "all_calls_avg_ns": sum(x[1] for x in rows) / len(rows)}, open(out, "w"))


if __name__ == "__main__":
    main(*sys.argv[1:4])
