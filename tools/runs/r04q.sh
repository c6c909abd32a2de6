#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${1:-r04q}
mkdir -p $O
cd $R
line() { echo "$1: $(grep -o '"ms_per_step": [0-9.]*' $2) $(grep -o '"tokens_per_s": [0-9.]*' $2) $(grep -o '"per_step": {[^}]*}' $2) $(grep -o '"ms_single_steps": [^]]*]' $2)"; }
f=$O/c16m_pf.json
timeout -k 10 400 python tools/c4_zipf_probe.py --mode cached --cache-rows 16000000 --stage-tokens 262144 --steps 20 --warmup 400 --prefetch-next > $f 2> ${f%.json}.err; line "probe 16M pf steps20" $f
f=$O/c16m_nopf.json
timeout -k 10 400 python tools/c4_zipf_probe.py --mode cached --cache-rows 16000000 --stage-tokens 262144 --steps 20 --warmup 400 > $f 2> ${f%.json}.err; line "probe 16M nopf" $f
timeout -k 10 500 python bench.py --quick --steps 5 --warmup 2 --no-hbm-variant 2>$O/bench_quick.err | tail -c 300
python - <<'PY'
import sys, time, torch
sys.path.insert(0, ".")
import bench
class A: pass
a = A(); a.pinned_rows=100_000_000; a.pinned_zipf_steps=20; a.pinned_zipf_warmup=400; a.pinned_cache_rows=16_000_000; a.pinned_stage_tokens=262144
t0=time.time()
n1, z = bench.pinned_baseline(a, lambda: torch.cuda.synchronize(), zipf_too=True)
print("pinned_baseline took", time.time()-t0)
print({k: z.get(k) for k in ("value","ms_per_step","rows_over_pcie_per_step","cache_hit_rate_of_distinct_cold_rows","error")}, z.get("zero_copy_same_stream"), z.get("zero_copy_static_head_same_hbm"))
PY
