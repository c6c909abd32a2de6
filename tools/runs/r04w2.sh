#!/bin/bash
# the self-launched 2-rank rehearsal of bench.py on one GPU (gloo), stderr kept whole
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${1:-r04w2}
mkdir -p $O
cd $R
env -u WORLD_SIZE -u RANK -u LOCAL_RANK -u MASTER_PORT SCONE_DIST_BACKEND=gloo SCONE_ONE_DEVICE=1 HSA_ENABLE_IPC_MODE_LEGACY=0 PYTHONFAULTHANDLER=1 \
  timeout -k 10 500 python bench.py --gpus 2 --steps 3 --warmup 1 --rows 200000 --batch 128 --sharded-rows-per-rank 200000 --sharded-steps 1 \
  --pinned-rows 200000 --cpu-seconds 1 > $O/out.txt 2> $O/err.txt
echo "rc=$?"
grep -n "Fatal\|Segmentation\|Thread 0x\|Current thread" $O/err.txt | head -20
