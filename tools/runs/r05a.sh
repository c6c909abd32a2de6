#!/bin/bash
# Round 5, first GPU call: the prefetched match (tests, A/B of the headline loop with / without the announcement, one batch
# repeated against a different batch every step, side-stream priority), and a kernel trace of the announced loop.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${1:-r05a}
mkdir -p $O
cd $R
timeout -k 10 900 python -m pytest tests/test_gpu_prefetch.py -x -q -m gpu > $O/pytest_prefetch.log 2>&1
echo "pytest prefetch rc=$? $(tail -3 $O/pytest_prefetch.log | tr '\n' ' ' | head -c 400)"
timeout -k 10 900 python -m pytest tests/test_gpu_bench_shape.py -x -q -m gpu -k "headline or cold_row or pinned" > $O/pytest_shape.log 2>&1
echo "pytest shape rc=$? $(tail -3 $O/pytest_shape.log | tr '\n' ' ' | head -c 400)"
ab() {  # name, env..., -- args
  local name=$1; shift
  timeout -k 10 300 env "$@" > $O/$name.json 2> $O/$name.err
  python3 - $O/$name.json $name <<'PY'
import json, sys
try:
    r = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    rf = r["roofline"]
    print("%-28s %6.3f G tok/s  step %7.4f ms  kernel %7.4f ms (min %7.4f med %7.4f)  step-kernel %6.1f us" % (
        sys.argv[2], r["value"] / 1e9, r["ms_per_step"], rf["avg_kernel_ms"], rf["kernel_ms"]["min"], rf["kernel_ms"]["median"],
        rf["step_minus_kernel_us"]), flush=True)
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
}
for rep in 1 2; do
ab rot_pf_$rep      X=1 python bench.py --quick --steps 40 --warmup 5
ab rot_serial_$rep  X=1 python bench.py --quick --steps 40 --warmup 5 --prefetch off
ab same_pf_$rep     X=1 python bench.py --quick --steps 40 --warmup 5 --same-batch
ab same_serial_$rep X=1 python bench.py --quick --steps 40 --warmup 5 --same-batch --prefetch off
ab rot_pf_prio0_$rep SCONE_PF_PRIORITY=0 python bench.py --quick --steps 40 --warmup 5
done
( cd /tmp && export TMPDIR=/tmp
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 $R/bench.py --quick --steps 20 --warmup 3 > $O/trace.log 2>&1 )
cp $(ls $O/trace/*/*kernel_stats.csv | head -1) $O/kernel_stats.csv
python3 - $O <<'PY'
import csv, glob, sys
O = sys.argv[1]
f = glob.glob(O + "/trace/*/*kernel_trace.csv")
if f:
    rows = list(csv.DictReader(open(f[0])))
    ks = [r for r in rows if "k_embed_wave" in r["Kernel_Name"] or "k_match_ell" in r["Kernel_Name"]]
    ks.sort(key=lambda r: int(r["Start_Timestamp"]))
    t0 = int(ks[-40]["Start_Timestamp"])
    with open(O + "/timeline_tail.txt", "w") as w:
        for r in ks[-40:]:
            w.write("%-14s q%-3s start %9.1f us  dur %8.1f us\n" % ("match" if "match" in r["Kernel_Name"] else "gather", r.get("Queue_Id", "?"),
                    (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
    print(open(O + "/timeline_tail.txt").read()[-1600:])
PY
rm -rf $O/trace
head -8 $O/kernel_stats.csv | cut -c1-160
