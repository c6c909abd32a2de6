#!/bin/bash
# Round 5, second GPU call: the announced match against the side stream's priority, the number of record buffers (2: the match
# can only start when the gather BEFORE the current one ends, i.e. together with the current one; 3: one gather earlier) and
# the release scope of the ordering events.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${1:-r05b}
mkdir -p $O
cd $R
timeout -k 10 600 python -m pytest tests/test_gpu_prefetch.py -x -q -m gpu > $O/pytest_prefetch.log 2>&1
echo "pytest prefetch rc=$? $(tail -3 $O/pytest_prefetch.log | tr '\n' ' ' | head -c 400)"
ab() {  # name, env..., -- args
  local name=$1; shift
  timeout -k 10 300 env "$@" > $O/$name.json 2> $O/$name.err
  python3 - $O/$name.json $name <<'PY'
import json, sys
try:
    r = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    rf = r["roofline"]
    print("%-34s %6.3f G tok/s  step %7.4f ms  kernel %7.4f ms (min %7.4f med %7.4f)  step-kernel %6.1f us" % (
        sys.argv[2], r["value"] / 1e9, r["ms_per_step"], rf["avg_kernel_ms"], rf["kernel_ms"]["min"], rf["kernel_ms"]["median"],
        rf["step_minus_kernel_us"]), flush=True)
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
}
A="python bench.py --quick --steps 40 --warmup 6"
for rep in 1 2; do
ab serial_$rep            X=1 $A --prefetch off
for prio in low normal high; do
for slots in 2 3 4; do
ab pf_${prio}_s${slots}_$rep  SCONE_PF_PRIORITY=$prio SCONE_PF_SLOTS=$slots $A
done; done
ab pf_low_s3_sysscope_$rep SCONE_PF_PRIORITY=low SCONE_PF_SLOTS=3 SCONE_PF_EVENT_SYSTEM_SCOPE=1 $A
done
( cd /tmp && export TMPDIR=/tmp
  SCONE_PF_PRIORITY=low SCONE_PF_SLOTS=3 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 $R/bench.py --quick --steps 20 --warmup 4 > $O/trace.log 2>&1 )
cp $(ls $O/trace/*/*kernel_stats.csv | head -1) $O/kernel_stats.csv
python3 - $O <<'PY'
import csv, glob, sys
O = sys.argv[1]
f = glob.glob(O + "/trace/*/*kernel_trace.csv")
if f:
    rows = list(csv.DictReader(open(f[0])))
    ks = [r for r in rows if "k_embed_wave" in r["Kernel_Name"] or "k_match_ell" in r["Kernel_Name"]]
    ks.sort(key=lambda r: int(r["Start_Timestamp"]))
    t0 = int(ks[-30]["Start_Timestamp"])
    with open(O + "/timeline_tail.txt", "w") as w:
        for r in ks[-30:]:
            w.write("%-8s q%-3s start %9.1f  end %9.1f  dur %8.1f us\n" % ("match" if "match" in r["Kernel_Name"] else "gather", r.get("Queue_Id", "?"),
                    (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
    print(open(O + "/timeline_tail.txt").read()[-1500:])
PY
rm -rf $O/trace
