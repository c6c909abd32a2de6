#!/bin/bash
# Round 6, first session: (1) the timing probes retaken on the current kernel (tools/build_variant.sh with the
# SCONE_PROBE_* switches of scone_embed_wave.h), two alternating rounds on one box; (2) the lockstep build (a workgroup
# barrier per token) -- time and bytes past L2; (3) the traffic probe of a token-ordered lookup (tools/run_order_probe.py).
#   builds (in the build container):  for v in no_wte rows_local no_store no_reads nothing lockstep perm: tools/build_variant.sh ...
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${1:-r06a}
mkdir -p $O
PART=${2:-all}
cd $R
if [ $PART = all ] || [ $PART = probes ]; then
python3 tools/mem_rates.py > $O/mem_rates.json 2> $O/mem_rates.err; cat $O/mem_rates.json
tools/ab_multi.sh ${1:-r06a}/probes 2 scone_amd/csrc/libscone_hip.so gpurun_ab/libno_wte.so gpurun_ab/librows_local.so \
  gpurun_ab/libno_store.so gpurun_ab/libno_reads.so gpurun_ab/libnothing.so gpurun_ab/liblockstep.so | tee $O/probes.txt
cd /tmp && export TMPDIR=/tmp
for v in normal lockstep; do
  if [ $v = normal ]; then unset SCONE_HIP_LIB; else export SCONE_HIP_LIB=$R/gpurun_ab/lib$v.so; fi
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout -k 10 200 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc_${v}_$c -- python3 $R/bench.py --steps 5 --warmup 2 --quick > $O/pmc_${v}_$c.log 2>&1
  done
done
fi
cd /tmp && export TMPDIR=/tmp
if [ $PART = all ] || [ $PART = order ]; then
export SCONE_HIP_LIB=$R/gpurun_ab/libperm.so
for stream in uniform zipf; do
  timeout -k 10 300 python3 $R/tools/run_order_probe.py --steps 12 --stream $stream > $O/run_order_${stream}_timing.json 2> $O/run_order_${stream}_timing.err
  tail -1 $O/run_order_${stream}_timing.json
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout -k 10 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc_order_${stream}_$c -- python3 $R/tools/run_order_probe.py --steps 6 --stream $stream > $O/run_order_${stream}_$c.json 2> $O/run_order_${stream}_$c.err
  done
done
fi
unset SCONE_HIP_LIB
python3 - $O <<'PY'
import csv, glob, json, sys, collections
O = sys.argv[1]
def per_launch(d, want="k_embed_wave"):
    f = glob.glob(d + "/*/*counter_collection.csv")
    if not f:
        return None
    rows = [(int(r.get("Dispatch_Id", 0)), float(r["Counter_Value"])) for r in csv.DictReader(open(f[0])) if want in r["Kernel_Name"]]
    rows.sort()
    return [v for _, v in rows]
res = {}
for v in ("normal", "lockstep"):
    e = {}
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        x = per_launch(f"{O}/pmc_{v}_{c}")
        if x:
            x = x[2:]                                   # warm-up launches
            e[c + "_KB_per_launch"] = sum(x) / len(x)
    if "FETCH_SIZE_KB_per_launch" in e:
        e["reads_past_L2_GB"] = 2 * e["FETCH_SIZE_KB_per_launch"] * 1024 / 1e9
    if "WRITE_SIZE_KB_per_launch" in e:
        e["writes_past_L2_GB"] = e["WRITE_SIZE_KB_per_launch"] * 1024 / 1e9
    res[v] = e
for stream in ("uniform", "zipf"):
    e = {}
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        x = per_launch(f"{O}/pmc_order_{stream}_{c}")
        if x and len(x) >= 3 + 18:
            x = x[3:]
            n = len(x) // 3
            for i, name in enumerate(("normal", "position_order", "token_order")):
                e.setdefault(name, {})[c + "_KB_per_launch"] = sum(x[i * n:(i + 1) * n]) / n
    for name, d in e.items():
        if "FETCH_SIZE_KB_per_launch" in d:
            d["reads_past_L2_GB"] = 2 * d["FETCH_SIZE_KB_per_launch"] * 1024 / 1e9
    try:
        e["timing"] = json.loads(open(f"{O}/run_order_{stream}_timing.json").read().strip().splitlines()[-1])
    except Exception as ex:
        e["timing_error"] = str(ex)
    res["run_order_" + stream] = e
json.dump(res, open(O + "/summary.json", "w"), indent=1)
print(json.dumps(res, indent=1)[:6000])
PY
rm -rf $O/pmc_*/*/*kernel_trace.csv $O/pmc_*/*/*agent_info.csv
