#!/bin/bash
# Round 5: the gather's read traffic past L2 with and without the bigram / trigram rows (tools/row_reread_probe.py).
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${1:-r05l}
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum"; do
  n=$(echo $c | tr " " "_" | cut -c1-24)
  timeout -k 10 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc_$n -- python3 $R/tools/row_reread_probe.py --steps 6 > $O/probe_$n.json 2> $O/probe_$n.err
done
python3 - $O <<'PY'
import csv, glob, json, sys, collections
O = sys.argv[1]
res = {}
for d in sorted(glob.glob(O + "/pmc_*/")):
    f = glob.glob(d + "*/*counter_collection.csv")
    if not f:
        continue
    rows = [r for r in csv.DictReader(open(f[0])) if "k_embed_wave" in r["Kernel_Name"]]
    by_counter = collections.defaultdict(list)
    for r in rows:
        by_counter[r["Counter_Name"]].append((int(r.get("Dispatch_Id", 0)), float(r["Counter_Value"])))
    for c, v in by_counter.items():
        v.sort()
        vals = [x for _, x in v]
        n = (len(vals) - 2) // 2
        res[c] = {"full": sum(vals[2:2 + n]) / n, "unigrams_only": sum(vals[2 + n:2 + 2 * n]) / n, "launches_each": n}
info = json.loads(open(glob.glob(O + "/probe_FETCH_SIZE.json")[0]).read().strip().splitlines()[-1])
out = {"what": "k_embed_wave, headline table and batches, per launch: full 1M-key index against an index of the unigrams only", "counters": res, **info}
if "FETCH_SIZE" in res:
    a, b = res["FETCH_SIZE"]["full"] * 2 * 1024, res["FETCH_SIZE"]["unigrams_only"] * 2 * 1024
    comp = info["full"]["distinct_rows_beyond_the_unigrams"] * 770
    out["reads_past_L2_GB"] = {"full": a / 1e9, "unigrams_only": b / 1e9, "bigram_trigram_rows": (a - b) / 1e9,
                               "bigram_trigram_rows_compulsory": comp / 1e9, "re_read_factor": (a - b) / comp}
json.dump(out, open(O + "/row_reread_probe.json", "w"), indent=1)
print(json.dumps(out, indent=1)[:2500])
PY
rm -rf $O/pmc_*/*/*kernel_trace.csv
