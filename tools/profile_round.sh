#!/bin/bash
# Collect one round's evidence on the GPU box (run through gpurun from the repo root):
#   tools/profile_round.sh <tag> [extra bench args]
# writes gpurun_out/<tag>/{bench.json, kernel_stats.csv, pmc_*.csv}; tools/summarize_profile.py
# turns them into profiles/<tag>/ (committed).
set -u
TAG=${1:-r00}; shift || true
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
timeout 900 python bench.py --steps 50 --warmup 5 "$@" > $O/bench.json 2> $O/bench.err
tail -c 400 $O/bench.err
cd /tmp && export TMPDIR=/tmp
# kernel trace and counters in separate runs (never --pmc together with other trace domains)
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 $R/bench.py --steps 20 --warmup 3 --quick "$@" > $O/trace.log 2>&1
for c in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum" \
         "TCC_EA0_RDREQ_DRAM_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_DRAM_sum TCC_EA0_WRREQ_64B_sum" \
         "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" \
         "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_SALU GRBM_GUI_ACTIVE"; do
  n=$(echo $c | tr " " "_" | cut -c1-32)
  timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc_$n -- python3 $R/bench.py --steps 5 --warmup 2 --quick "$@" > $O/pmc_$n.log 2>&1
done
cp $(ls $O/trace/*/*kernel_stats.csv | head -1) $O/kernel_stats.csv
python3 $R/tools/timed_kernel_avg.py $O/trace 20 $O/kernel_timed.json   # the 20 timed launches alone (not warm-up / output-buffer trials)
cat $O/bench.json
