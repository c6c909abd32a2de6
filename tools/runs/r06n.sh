#!/bin/bash
# Round 6: the output buffer's placement effect against the vector L1's address-translation counters (tools/out_tlb_probe.py).
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${1:-r06n}
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
python3 $R/tools/out_tlb_probe.py > $O/plain.json 2> $O/plain.err; tail -1 $O/plain.json
for c in "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_REQUEST_sum" "TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_STALL_UTCL2_REQ_OUT_OF_CREDITS" "FETCH_SIZE" "WRITE_SIZE" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_WRREQ_STALL_sum"; do
  n=$(echo $c | tr " " "_" | cut -c1-40)
  timeout -k 10 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc_$n -- python3 $R/tools/out_tlb_probe.py > $O/run_$n.json 2> $O/run_$n.err
  tail -1 $O/run_$n.json
done
python3 - $O <<'PY'
import csv, glob, json, sys, collections
O = sys.argv[1]
out = {}
for d in sorted(glob.glob(O + "/pmc_*/")):
    f = glob.glob(d + "*/*counter_collection.csv")
    if not f:
        continue
    by = collections.defaultdict(list)
    for r in csv.DictReader(open(f[0])):
        if "k_embed_wave" in r["Kernel_Name"]:
            by[r["Counter_Name"]].append((int(r.get("Dispatch_Id", 0)), float(r["Counter_Value"])))
    tag = d.rstrip("/").split("pmc_")[-1]
    try:
        times = json.loads(open(f"{O}/run_{tag}.json").read().strip().splitlines()[-1])["kernel_ms_per_buffer"]
    except Exception:
        times = None
    for c, v in by.items():
        v.sort()
        vals = [x for _, x in v]
        per = [sum(vals[i * 6 + 2:i * 6 + 6]) / 4 for i in range(len(vals) // 6)]
        out[c] = {"per_buffer": per, "kernel_ms_per_buffer_in_that_run": times}
json.dump(out, open(O + "/summary.json", "w"), indent=1)
for c, e in out.items():
    print(c, [round(x) for x in e["per_buffer"]], [round(t, 4) for t in (e["kernel_ms_per_buffer_in_that_run"] or [])])
PY
rm -rf $O/pmc_*/*/*kernel_trace.csv $O/pmc_*/*/*agent_info.csv
