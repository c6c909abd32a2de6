#!/bin/bash
# Round 5: bytes past L2 of the sliding-window lookup against k_embed_wave (FETCH_SIZE / WRITE_SIZE / TCC passes of bench.py
# --quick with SCONE_SLIDE=0 / 1, same box) + the in-process A/B.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${1:-r05n}
mkdir -p $O
cd $R
timeout -k 10 300 python tools/slide_ab.py --segs 16,32 --rounds 2 > $O/slide_ab.json 2> $O/slide_ab.err; grep -E "^wave|^slide" $O/slide_ab.err | cut -c1-110
cd /tmp && export TMPDIR=/tmp
for sl in 0 1; do
  for c in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum" "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_SMEM"; do
    n=$(echo $c | tr " " "_" | cut -c1-20)
    SCONE_SLIDE=$sl SCONE_SLIDE_SEG=32 timeout -k 10 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/s${sl}_pmc_$n -- python3 $R/bench.py --steps 5 --warmup 2 --quick > $O/s${sl}_pmc_$n.log 2>&1
  done
done
python3 - $O <<'PY'
import csv, glob, json, sys, collections
O = sys.argv[1]
res = {}
for sl in ("0", "1"):
    agg = collections.defaultdict(list)
    for d in glob.glob(O + f"/s{sl}_pmc_*/"):
        for f in glob.glob(d + "*/*counter_collection.csv"):
            for r in csv.DictReader(open(f)):
                k = r["Kernel_Name"]
                short = next((s for s in ("k_embed_slide", "k_embed_wave", "k_match_win", "k_match_ell") if s in k), None)
                if short:
                    agg[(short, r["Counter_Name"])].append(float(r["Counter_Value"]))
    for (k, c), v in agg.items():
        res.setdefault(k, {})[c] = sum(v) / len(v)
json.dump(res, open(O + "/pmc_slide_vs_wave.json", "w"), indent=1, sort_keys=True)
for k, v in res.items():
    if "FETCH_SIZE" in v:
        print(k, "reads past L2 %.3f GB, writes %.3f GB" % (2 * v["FETCH_SIZE"] * 1024 / 1e9, v.get("WRITE_SIZE", 0) * 1024 / 1e9), {c: round(x) for c, x in v.items() if c.startswith(("TCC", "SQ_INSTS", "SQ_WAVES"))})
PY
rm -rf $O/s*_pmc_*/*/*kernel_trace.csv
