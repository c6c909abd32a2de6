set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r05o
mkdir -p $O
cd $R
for rep in 1 2 3; do
  for v in normal probe; do
    if [ $v = probe ]; then export SCONE_HIP_LIB=$R/gpurun_ab/libprobe_scales_local.so; else unset SCONE_HIP_LIB; fi
    timeout -k 10 200 python bench.py --quick --steps 40 --warmup 5 > $O/${v}_$rep.json 2> $O/${v}_$rep.err
    python3 - $O/${v}_$rep.json $v$rep <<'PY'
import json, sys
r = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); rf = r["roofline"]
print("%-8s step %.4f kernel %.4f (min %.4f med %.4f)" % (sys.argv[2], r["ms_per_step"], rf["avg_kernel_ms"], rf["kernel_ms"]["min"], rf["kernel_ms"]["median"]), flush=True)
PY
  done
done
cd /tmp && export TMPDIR=/tmp
export SCONE_HIP_LIB=$R/gpurun_ab/libprobe_scales_local.so
timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_probe -- python3 $R/bench.py --steps 5 --warmup 2 --quick > $O/pmc_probe.log 2>&1
python3 - $O <<'PY'
import csv, glob, sys
v = [float(r["Counter_Value"]) for f in glob.glob(sys.argv[1] + "/pmc_probe/*/*counter_collection.csv") for r in csv.DictReader(open(f)) if "k_embed_wave" in r["Kernel_Name"]]
print("probe: reads past L2 %.3f GB per launch (normal: 2.00)" % (2 * sum(v) / len(v) * 1024 / 1e9))
PY
