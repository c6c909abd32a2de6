#!/bin/bash
# One bench line per BASELINE.json config that fits one GPU (C2, C3, C4 in HBM and in pinned host DRAM, shard 0/8 of
# C5's table), same box, appended to gpurun_out/<tag>/configs.jsonl.   tools/run_configs.sh <tag>
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${1:-configs}
mkdir -p $O
cd $R
run() {
  name=$1; shift
  echo "== $name: bench.py $*" | tee -a $O/configs.log
  timeout -k 10 900 python bench.py --quick "$@" > $O/$name.json 2>> $O/configs.log || { echo "$name FAILED"; tail -5 $O/configs.log; return 1; }
  python3 - $O/$name.json $name >> $O/configs.jsonl <<'PY'
import json, sys
r = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r["config_name"] = sys.argv[2]
print(json.dumps(r))
PY
  python3 - $O/$name.json $name <<'PY'
import json, sys
r = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
rf = r["roofline"]
print("%-38s %6.3f G tok/s  step %6.3f ms  kernel %6.3f ms  frac %.3f (%s)  algorithmic %.3f" % (
    sys.argv[2], r["value"] / 1e9, r["ms_per_step"], rf["avg_kernel_ms"], rf["frac"],
    "left L2" if rf.get("traffic") is not None else "compulsory", rf.get("algorithmic_frac", 0.0)), flush=True)
PY
}
rm -f $O/configs.jsonl
run headline_int8_1M_d768 --steps 50 --warmup 5 || exit 1
run headline_zipf_stream --steps 50 --warmup 5 --stream zipf || exit 1
run C2_fp16_1M_d768 --steps 50 --warmup 5 --format fp16 || exit 1
run C3_int8_10M_d1024 --steps 30 --warmup 3 --rows 10000000 --dim 1024 --keygen zipf_gpu || exit 1
run int4_1M_d1024 --steps 30 --warmup 3 --format int4 --dim 1024 || exit 1
run int8_1M_d1280 --steps 30 --warmup 3 --dim 1280 || exit 1
run C4_int4_100M_d1024_hbm --steps 20 --warmup 3 --rows 100000000 --format int4 --dim 1024 --keygen structured || exit 1
run C4_int4_100M_d1024_pinned --steps 10 --warmup 2 --rows 100000000 --format int4 --dim 1024 --keygen structured --placement pinned_host --hot-rows 1000000 --stage-tokens 131072 || exit 1
run C4_int4_100M_d1024_pinned_zero_copy --steps 10 --warmup 2 --rows 100000000 --format int4 --dim 1024 --keygen structured --placement pinned_host --hot-rows 1000000 --stage-tokens 0 || exit 1
run C5_shard0of8_int4_1B_d1024 --steps 10 --warmup 2 --rows 1000000000 --format int4 --dim 1024 --keygen structured --shard-of 0/8 || exit 1
